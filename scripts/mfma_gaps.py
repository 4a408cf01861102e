"""Instruction mix between consecutive matrix instructions in the largest loop of a kernel (from hipcc -S output).
usage: python scripts/mfma_gaps.py file.s mangled_prefix"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name)
i = s.index('\n', s.index(name + '', i))
j = s.index('.Lfunc_end', i)
body = [l for l in s[i:j].split('\n') if l.strip() and not l.strip().startswith(';')]
lab = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: lab[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in lab and lab[t] < n: loops.append((n - lab[t], lab[t], n))
loops.sort(reverse=True)
# the innermost loop that still holds matrix instructions: the smallest one with at least 48 of them
cands = [(L, a, b) for (L, a, b) in loops if sum(1 for l in body[a:b] if l.strip().startswith('v_mfma')) >= 48]
L, a, b = min(cands) if cands else loops[0]
gaps, cur, kinds = [], 0, Counter()
tot = Counter()
for l in body[a:b]:
    op = l.strip().split()[0]
    if op.startswith('v_mfma'):
        gaps.append(cur); cur = 0; tot['mfma'] += 1
    else:
        cur += 1
        k = 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'scratch_')) else 'wait' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop') else 'salu'
        tot[k] += 1
print('loop lines', L, dict(tot))
print('gap histogram (other instructions before a matrix instruction):', sorted(Counter(gaps).items()))
print('largest gaps:', sorted(gaps, reverse=True)[:12])
