#!/usr/bin/env python3
"""Instruction mix of the longest loop of a kernel in hipcc's assembly output.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s file.hip -I pde_amd/csrc
    python scripts/count_loop.py k.s <substring of the mangled kernel name>
"""
import collections
import re
import sys


def main(path, pat):
    txt = open(path).read()
    names = [n for n in re.findall(r'^(_Z\w+):', txt, re.M) if pat in n]
    for name in names:
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
        if not loops:
            print(name, "no loop")
            continue
        lo, hi = max(loops, key=lambda t: t[1] - t[0])
        c = collections.Counter()
        for l in body[lo:hi]:
            l = l.strip()
            if not l or l.startswith(('.', ';')):
                continue
            op = l.split()[0]
            if op.startswith('v_mfma'):
                c['MFMA'] += 1
            elif op.startswith('v_'):
                c['VALU'] += 1
                c[op] += 1
            elif op.startswith('ds_'):
                c['DS'] += 1
                c[op] += 1
            elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                c['VMEM'] += 1
                c[op] += 1
            elif op.startswith('s_waitcnt'):
                c['s_waitcnt'] += 1
            elif op.startswith('s_'):
                c['SALU'] += 1
        print(name, "loop lines", hi - lo)
        print("  " + "  ".join(f"{k}={v}" for k, v in c.most_common(60)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
