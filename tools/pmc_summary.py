#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv rows per kernel and counter.
    python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b ... [--match substr]
"""
import collections
import csv
import glob
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for d in args:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if match and match not in k:
                    continue
                key = (k[:70], row["Counter_Name"])
                acc[key][0] += float(row["Counter_Value"])
                acc[key][1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        print(f"{k:70s} {c:22s} avg {s / n:16.1f}  over {n} dispatches")


if __name__ == "__main__":
    main()
