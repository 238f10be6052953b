#!/usr/bin/env python3
"""Instruction mix of the hot loop of each kernel in a gfx950 assembly file (hipcc -S --cuda-device-only).

For every function the largest backward-branch loop is taken as the hot loop; instructions are
grouped by unit.  Usage: isa_mix.py file.s [substring of the kernel name]"""
import collections
import re
import sys


def classify(m):
    if m.startswith(("v_pk_",)):
        return "valu_pk"
    if m.startswith("v_") and ("dpp" in m):
        return "valu_dpp"
    if m.startswith(("v_mov", "v_accvgpr", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_move"
    if m.startswith("v_"):
        return "valu_other"
    if m.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_ld"
    if m.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_st"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    if m.startswith("s_nop"):
        return "nop"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    funcs, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur and ln.startswith("\t.end_amdhsa_kernel") or (cur and ln.startswith(".Lfunc_end")):
            cur = None
            continue
        if cur is not None:
            funcs[cur].append(ln)
    for name, body in funcs.items():
        if want not in name:
            continue
        labels = {}
        insts = []
        for ln in body:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            s = ln.strip()
            if not s or s.startswith((";", ".", "//")):
                continue
            insts.append(s)
        best = None
        for i, s in enumerate(insts):
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", s)
            if m:
                tgt = m.group(1) or m.group(2)
                if tgt in labels and labels[tgt] <= i:
                    span = (labels[tgt], i)
                    if best is None or span[1] - span[0] > best[1] - best[0]:
                        best = span
        print(name)
        print("  total instructions:", len(insts), " hot loop:", best, "=", (best[1] - best[0] + 1) if best else 0)
        if not best:
            continue
        mix = collections.Counter()
        detail = collections.Counter()
        dpp = 0
        for s in insts[best[0]:best[1] + 1]:
            mn = s.split()[0]
            c = classify(mn)
            if c.startswith("valu") and ("row_" in s or "wave_" in s or "quad_perm" in s or "dpp" in s):
                c = "valu_dpp"
            mix[c] += 1
            if c.startswith("valu"):
                detail[mn] += 1
        valu = sum(v for k, v in mix.items() if k.startswith("valu"))
        print("  ", dict(sorted(mix.items())), " VALU total:", valu)
        print("  ", detail.most_common(24))


if __name__ == "__main__":
    main()
