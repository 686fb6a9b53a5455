#!/usr/bin/env python3
"""Static instruction mix of one kernel from a hipcc -save-temps .s file, per basic block.

usage: isa_mix.py file.s kernel_symbol_substring [min_block_size]

For each basic block (label to label) prints counts per class and an issue-cycle estimate
(MFMA 16x16x4 f32 = 32 cycles, other VALU = 4, v_pk = 4, LDS/VMEM/scalar = 1 issue slot), plus the
branch targets at the block's end so loops can be recognised.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accmov"
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_mov") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "vmov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_add") or op.startswith("ds_pk_add"):
        return "ds_atomic"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
        return "ds_perm"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "ds_write"
    if op.startswith("ds_"):
        return "ds_other"
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "vm_load"
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("flat_store") or op.startswith("scratch_store"):
        return "vm_store"
    if op.startswith("buffer_atomic") or op.startswith("global_atomic"):
        return "vm_atomic"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


COST = {"mfma": 32, "valu": 4, "vpk": 4, "vmov": 4, "accmov": 4}


def main():
    path, sym = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if ln.endswith(":") or ": ;" in ln:
            name = ln.split(":")[0]
            if sym in name and not name.startswith(".") and not name.startswith(";"):
                start = i
                break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    tot = Counter()
    print(f"{'block':>12} {'n':>6} {'cyc':>7}  mix / branches")
    for name, ins in blocks.items():
        c = Counter(classify(op) for op, _ in ins)
        tot.update(c)
        cyc = sum(COST.get(k, 1) * v for k, v in c.items())
        if len(ins) >= minsz:
            br = [s.split()[-1] for op, s in ins if op.startswith("s_cbranch") or op.startswith("s_branch")]
            mix = " ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]))
            print(f"{name:>12} {len(ins):6d} {cyc:7d}  {mix}  -> {','.join(br)}")
    print("total static:", dict(tot))


if __name__ == "__main__":
    main()
