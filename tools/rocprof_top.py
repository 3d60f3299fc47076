#!/usr/bin/env python3
"""Prints the per-kernel summary (calls, total, average us) of a rocprofv3 --kernel-trace results .db."""
import glob
import sqlite3
import sys

for path in sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True) + glob.glob(sys.argv[1] + "/*_results.db")):
    c = sqlite3.connect(path)
    print(path)
    print(f"{'kernel':90s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s}")
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        print(f"{name[:90]:90s} {calls:6d} {avg:9.3f} {total / 1e3:9.3f}")
    break
