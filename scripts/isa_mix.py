#!/usr/bin/env python3
"""Instruction mix of a kernel's basic blocks from hipcc -S device assembly (no GPU needed).

    python scripts/isa_mix.py <file.s> <kernel-name-substring> [min_block_instructions]

Prints every basic block with its instruction-class counts; loops show up as blocks whose terminating
branch targets an earlier (or the same) label.  VALU issue is what bounds the compositing kernels
(DESIGN.md section 4), so "valu per trip" is their cost model.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op: str) -> str:
    if op.startswith("v_"):
        if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
            return "valu_trans"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if op.startswith("v_cmp"):
            return "valu_cmp"
        if op.startswith("v_cndmask"):
            return "valu_cndmask"
        if op.startswith("v_permlane") or "_dpp" in op:
            return "valu_xlane"
        if op.startswith("v_pk_"):
            return "valu_pk"
        return "valu"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            return "s_wait"
        if op.startswith(("s_cbranch", "s_branch")):
            return "s_branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and needle in m.group(1):
            start = i
            print("kernel", m.group(1))
            break
    assert start is not None, "kernel not found"
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or re.match(r"^\.Lfunc_end", l):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        m = re.match(r"^\t([a-z_0-9]+)(\s|$)", l)
        if m and not l.startswith("\t."):
            blocks[cur].append((m.group(1), l.strip()))
    order = list(blocks)
    total = Counter()
    for name, ins in blocks.items():
        c = Counter(classify(op) for op, _ in ins)
        total.update(c)
        if len(ins) < min_n:
            continue
        back = ""
        for op, text in ins:
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = text.split()[-1]
                if tgt in blocks and order.index(tgt) <= order.index(name):
                    back = f"  <- loop back to {tgt}"
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        detail = " ".join(f"{k}={v}" for k, v in sorted(c.items()))
        print(f"{name:12s} n={len(ins):4d} VALU={valu:4d}  {detail}{back}")
    valu = sum(v for k, v in total.items() if k.startswith("valu"))
    print("whole kernel:", sum(total.values()), "instructions, VALU", valu, dict(total))


if __name__ == "__main__":
    main()
