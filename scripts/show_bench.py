"""Print the headline and the per-kernel split of a bench.py JSON line (file argument)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"{d['ms_per_step']:.3f} ms/step  {d['value'] / 1e6:.2f} M {d['unit']}  roofline {d['roofline']['kernel']} "
      f"{d['roofline']['achieved']:.1f}/{d['roofline']['peak']} {d['roofline']['unit']} frac {d['roofline']['frac']:.3f}")
for name, v in d["kernels"].items():
    print(f"  {name:28s} {v['ms_per_step']:.4f}")
