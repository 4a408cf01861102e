#!/usr/bin/env python3
"""Condensed trace of a kernel's longest loop in hipcc's assembly: memory operations and waits in program order,
with the number of matrix / vector instructions between them.

    python scripts/loop_trace.py k.s <substring of the mangled kernel name> [innermost]
"""
import re
import sys


def main(path, pat, which="longest"):
    txt = open(path).read()
    names = [n for n in re.findall(r'^(_Z\w+):', txt, re.M) if pat in n]
    name = names[0]
    start = txt.index(name + ":")
    end = txt.index("s_endpgm", start)
    body = txt[start:end].split("\n")
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    loops.sort(key=lambda t: t[1] - t[0])
    lo, hi = loops[-1] if which == "longest" else loops[int(which)]
    print(name, "loops:", [(a, b, b - a) for a, b in loops], "showing", lo, hi)
    nm = nv = 0
    for l in body[lo:hi]:
        l = l.strip()
        if not l or l.startswith(('.', ';')):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'):
            nm += 1
        elif op.startswith('v_'):
            nv += 1
        elif op.startswith(('ds_', 'global_', 'buffer_', 's_waitcnt', 's_barrier', 's_cbranch', 's_branch', 'scratch_')):
            print(f"  [{nm:3d} mfma {nv:3d} valu]  {l.split(';')[0].strip()[:90]}")
            nm = nv = 0
    print(f"  [{nm:3d} mfma {nv:3d} valu]  (end)")


if __name__ == "__main__":
    main(*sys.argv[1:])
