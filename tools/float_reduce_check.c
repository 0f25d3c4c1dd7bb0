// Exhaustive check of the float-reciprocal reduction gf_reduce_u32 (kosk_limb_dev.hpp) mod 3329 for every 32-bit x:  gcc -O2 tools/float_reduce_check.c -o /tmp/frc && /tmp/frc
#include <stdio.h>
#include <stdint.h>
int main(){
  const uint32_t Q=3329; const float c = 0x1.3afb72p-12f; /* = (float)((1 - 2^-22) / 3329): the constant of gf_reduce_u32 */
  uint64_t bad=0; uint32_t maxr=0;
  for (uint64_t x=0; x < (1ull<<32); x++){
    float f=(float)(uint32_t)x; uint32_t t=(uint32_t)(f*c);
    uint32_t r=(uint32_t)x - t*Q; if (r>maxr) maxr=r;
    uint32_t m = r < r-Q ? r : r-Q;
    if (m != (uint32_t)x%Q) { if(bad<5) printf("bad x=%llu t=%u r=%u\n",(unsigned long long)x,t,r); bad++; }
  }
  printf("c=%.10g bad=%llu maxr=%u\n", c, (unsigned long long)bad, maxr);
}
