#!/usr/bin/env python3
"""Static opcode-class histogram of one kernel of a hipcc -save-temps .s listing: what the non-FMA VALU
instructions ARE (40 % of the 7kdv kernel's non-MFMA VALU instructions are not FMAs).

usage: isa_classes.py file.s kernel_symbol_substring

Static counts (every instruction once; the kernel's layer loops run 3x for a 4-hidden-layer net, so the dynamic
mix leans a little further towards the loop bodies).  Classes:
  fma      v_fma_f32 / v_fmac_f32 / v_fma_f64 / v_pk_fma_f32
  mul add  v_mul_f32, v_add_f32, v_sub_f32 (and f64 forms)
  dpp      any VALU op with a DPP modifier (the 16-lane row sums: 4 moves/adds per sum)
  mov      v_mov_b32, v_accvgpr_*, v_readfirstlane, v_readlane, v_writelane  (copies, AGPR parking, spill lanes)
  select   v_cndmask, v_cmp*
  int      integer / bit / address arithmetic
  cvt      conversions (precise top: f32 <-> f64)
  trans    v_rcp, v_exp, v_sqrt, v_ldexp ...
"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
sym = sys.argv[2]
m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(sym), text, re.M)
if not m:
    sys.exit("kernel not found")
body = text[m.end():]
body = body[:body.index("s_endpgm")]
cls = Counter()
ops = Counter()
for ln in body.split("\n"):
    ln = ln.split(";")[0].strip()
    if not ln or ln.endswith(":") or ln.startswith("."):
        continue
    op = ln.split()[0]
    ops[op] += 1
    if op.startswith("v_mfma"):
        c = "mfma"
    elif not op.startswith("v_"):
        c = "s_nop" if op == "s_nop" else ("s_waitcnt" if op.startswith("s_waitcnt") else
             "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem")
    elif "dpp" in ln or "row_" in ln or "quad_perm" in ln:
        c = "valu dpp"
    elif re.match(r"v_(pk_)?(fma|fmac|mad|mac)_f(32|64)", op):
        c = "valu fma"
    elif re.match(r"v_(pk_)?(mul|add|sub|subrev)_f(32|64)", op):
        c = "valu mul/add"
    elif op.startswith(("v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane", "v_swap")):
        c = "valu mov"
    elif op.startswith(("v_cndmask", "v_cmp")):
        c = "valu select"
    elif op.startswith("v_cvt"):
        c = "valu cvt"
    elif op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_ldexp", "v_rndne", "v_frexp")):
        c = "valu trans/round"
    elif re.match(r"v_(max|min|med3)", op):
        c = "valu min/max"
    else:
        c = "valu int/bit"
    cls[c] += 1
tot = sum(cls.values())
valu = sum(v for k, v in cls.items() if k.startswith("valu"))
print("kernel %s: %d instructions, %d non-MFMA VALU, %d MFMA" % (m.group(1)[:70], tot, valu, cls["mfma"]))
for k, v in cls.most_common():
    print("  %-18s %6d  %5.1f %% of all%s" % (k, v, 100.0 * v / tot, ("  %5.1f %% of non-MFMA VALU" % (100.0 * v / valu)) if k.startswith("valu") else ""))
print("most frequent VALU opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(40) if kv[0].startswith("v_") and not kv[0].startswith("v_mfma")))
