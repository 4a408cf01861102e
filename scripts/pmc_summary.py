#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per kernel and counter, the mean over dispatches.

    python scripts/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_grbm ...
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void pdekan::", "").replace("pdekan::", "")


def main(dirs):
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            with open(path) as f:
                for row in csv.DictReader(f):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur = defaultdict(list)
            kt = path.replace("_counter_collection.csv", "_kernel_trace.csv")
            if os.path.exists(kt):
                with open(kt) as f:
                    for row in csv.DictReader(f):
                        dur[short(row["Kernel_Name"])].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
            print(f"# {path}")
            for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0]))):
                if "pdekan" not in path and not acc[k]:
                    continue
                if not any(s in k for s in ("mfma", "bf16x3", "f16", "head", "generic", "reduce", "sum_partials", "sparse", "pack", "absmax")):
                    continue
                n = len(next(iter(acc[k].values())))
                ms = (sum(dur[k]) / len(dur[k]) / 1e6) if dur.get(k) else float("nan")
                vals = " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items()))
                print(f"{k[:60]:60s} n={n:3d} avg_ms={ms:8.4f} {vals}")


if __name__ == "__main__":
    main(sys.argv[1:])
