#!/usr/bin/env python3
"""Instruction histogram of one kernel from a `hipcc --save-temps` assembly listing (.s): per basic block and for the hot
loop (the block range between two labels), opcode classes the accumulate-kernel write-up needs (multiply-adds vs glue).
usage: isa_hist.py file.s <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\S*:", l) and key in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    assert start is not None, "kernel not found"
    body = lines[start:end]
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = collections.Counter()
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
        if m and not l.strip().startswith((".", ";")):
            blocks[cur][m.group(1)] += 1
    total = collections.Counter()
    for b in blocks.values():
        total.update(b)
    meta = {}
    for l in lines[end:end + 80]:
        m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    print("meta:", meta)
    if show_blocks:
        for name, b in blocks.items():
            n = sum(b.values())
            if n >= 50:
                print(f"{name}: {n} instr, mad64 {b['v_mad_u64_u32'] + b['v_mad_i64_i32']}")
    # the hot block(s): everything with >= 1000 instructions
    hot = collections.Counter()
    for name, b in blocks.items():
        if sum(b.values()) >= 1000:
            hot.update(b)
    for title, c in (("whole kernel", total), ("blocks >= 1000 instructions (the addition)", hot)):
        n = sum(c.values())
        print(f"\n== {title}: {n} instructions")
        for op, k in c.most_common(40):
            print(f"  {op:28s} {k:6d}  {100.0 * k / n:5.1f} %")


if __name__ == "__main__":
    main()
