#!/usr/bin/env python3
"""Per-instruction budget of k_ntt256<PLAIN> (round 6, review item 6): compiles csrc/kosk_kernels.hip for gfx950, takes the kernel's ISA
and counts its instructions by what they are for.  No GPU needed.   python tools/ntt_budget.py [out.txt]"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    subprocess.check_call(["hipcc", "-O3", "-std=c++20", "-fPIC", "--offload-arch=gfx950", "-x", "hip", "-c",
                           os.path.join(ROOT, "mpcith_kyber_kosk_amd", "csrc", "kosk_kernels.hip"), "-o", os.path.join(td, "k.o"), "--save-temps=obj"], cwd=td,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open([os.path.join(td, f) for f in os.listdir(td) if "gfx950" in f and f.endswith(".s")][0]).read()
m = re.search(r"^_ZN4kosk8k_ntt256ILb1EEEvNS_7NttArgsE:(.*?)s_endpgm", asm, re.S | re.M)
body = m.group(1)
ops = [l.split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.rstrip().endswith(":")]
cnt = collections.Counter(ops)
vg = int(re.search(r"\.name:\s+_ZN4kosk8k_ntt256ILb1EEEvNS_7NttArgsE.*?\.vgpr_count:\s+(\d+)", asm, re.S).group(1))

def take(*names):
    return sum(cnt.pop(n, 0) for n in list(cnt) if n in names or any(n.startswith(p[:-1]) for p in names if p.endswith("*")))
bfly_mad, bfly_dot = cnt["v_mad_u16"], cnt["v_dot2_i32_i16"]
sdwa = take("v_sub_u32_sdwa", "v_add_u32_sdwa")
nb = sdwa // 2                                   # butterflies per thread: 7 layers x 8
fin_md = (bfly_mad - nb) + (bfly_dot - nb)       # the final Montgomery step by 2^16 mod q: one mad + one dot2 per coefficient
cnt.pop("v_mad_u16"); cnt.pop("v_dot2_i32_i16")
fin_pk = take("v_perm_b32", "v_pk_ashrrev_i16", "v_pk_add_u16", "v_pk_sub_i16", "v_and_b32")
lds = take("ds_*")
glob = take("global_*")
unpack = take("v_lshrrev_b32_e32")
salu = take("s_mov_b32", "s_movk_i32", "s_add_u32", "s_addc_u32", "s_getpc_b64", "s_and_b64", "s_or_b64", "s_lshl_b32", "s_lshl_b64", "s_sub_i32", "s_load_*", "s_and_saveexec_b64", "s_cbranch_*", "s_endpgm")
nops = take("s_nop")
waits = take("s_waitcnt", "s_barrier")
other_v = sum(v for k, v in cnt.items() if k.startswith("v_"))
rest = {k: v for k, v in cnt.items() if not k.startswith("v_")}
valu_static = 4 * nb + fin_md + fin_pk + unpack + other_v
valu_graded = valu_static  # the centred-output branch (kosk_ntt256_batch) runs every packed instruction of the epilogue
lines = ["k_ntt256<PLAIN> (kosk_ntt256_batch: 65 536 polynomials = 16 384 waves, 16 lanes x 16 coefficients per polynomial), ISA of the tree's hipcc build; %d VGPRs" % vg,
         "per thread (16 coefficients), static counts:",
         "  butterflies                 %3d x 4 = %3d   v_mad_u16 (op_sel) + v_dot2_i32_i16 + v_sub_u32_sdwa + v_add_u32_sdwa   (7 layers x 8)" % (nb, 4 * nb),
         "  final Montgomery step            %3d        one v_mad_u16 + one v_dot2_i32_i16 per coefficient (x -> x mod q in (-q, q))" % fin_md,
         "  packed canonicalisation          %3d        v_perm + (ashr, and, add) per pair -> [0, q); + (sub, ashr, and, sub) per pair -> centred (poly_reduce's range:" % fin_pk,
         "                                              what kosk_ntt256_batch asks for; the pipeline's canonical outputs skip those four: %d)" % (fin_pk - 32),
         "  unpack of the second stage        %3d        w >> 16 for the odd coefficients of the eight dwords read back from LDS" % unpack,
         "  addressing, predicates, moves     %3d" % other_v,
         "  = vector instructions            %3d        (of which butterflies %d %%)" % (valu_static, round(400.0 * nb / valu_static)),
         "  LDS instructions                  %3d        2 x ds_write_b128 + 16 x ds_read_i16 (stage in), 16 x ds_write_b16 + 2 x ds_read_b128 (the one transposition)" % lds,
         "  global loads / stores             %3d        2 + 2 x 16 bytes of data, the per-lane zeta pairs of layers 8, 4, 2" % glob,
         "  scalar instructions               %3d        (zeta literals of the uniform layers as s_mov, address arithmetic, branches), s_nop %d, waits / barriers %d%s" % (salu, nops, waits, (", other %r" % rest) if rest else ""),
         "",
         "Where the time is (round 4's SQ counters of this kernel, profiles/r04_ntt.txt: vector issue 75-80 %% busy at 4.5 cycles per wave64 instruction):",
         "  16 384 waves x %d vector instructions / 1 024 SIMDs = %d instructions per SIMD x 4.5 cycles = %.1f k cycles = %.1f us at 2.1 GHz of vector issue" % (
             valu_graded, 16 * valu_graded, 16 * valu_graded * 4.5 / 1e3, 16 * valu_graded * 4.5 / 2.1e3),
         "  against 19.4-20.0 us measured and 10.4 us for a plain copy of the same 67 MB: the kernel is the SUM of an issue-bound transform (the four",
         "  butterfly opcodes are 16-bit / SDWA / dot forms that issue every ~4.5 cycles, three times slower than a plain 32-bit v_add or v_bitop3) and a",
         "  memory phase it only partly overlaps (eight waves per SIMD, a workgroup's load -> transform -> store phases interleave across workgroups).",
         "Floor of this formulation: butterflies alone %d x 16 x 4.5 cycles = %.1f us; everything but the butterflies %d instructions = %.1f us; the review's" % (
             4 * nb, 16 * 4 * nb * 4.5 / 2.1e3, valu_graded - 4 * nb, 16 * (valu_graded - 4 * nb) * 4.5 / 2.1e3),
         "  0.50 of HBM peak (16.7 us) needs ~60 fewer vector instructions per thread (-17 %): the epilogue's 32 centring instructions are the only block that is",
         "  not arithmetic the transform needs (the pipeline's own launches, canonical outputs, already skip them); fusing the final reduction into the last layer",
         "  saves nothing (the Montgomery step is per coefficient either way: counted in DESIGN.md 16.9); a packed-fp32 butterfly (3 instead of 4 instructions, full rate)",
         "  was built in round 2 and is wrong beside int8-MFMA waves unless every operand is pinned to a VGPR, where it is no faster (DESIGN.md 9; removed in round 6)."]
out = "\n".join(lines) + "\n"
print(out)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(out)
