// placeholder: verifier pipeline lands in the next commit
#include "kosk_ctx.hpp"
namespace kosk {
int stage_verifier_inputs(Ctx &c, int, const uint8_t *, const uint8_t *) { c.err = "verifier not built yet"; return -1; }
int verify_resident(Ctx &c, int, uint8_t *) { c.err = "verifier not built yet"; return -1; }
}
