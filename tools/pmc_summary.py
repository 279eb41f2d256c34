#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv rows per (kernel, launch grid, counter). Usage: pmc_summary.py <dir> [substr]"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        if sub in name:
            agg[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, grid, c), v in sorted(agg.items()):
    print("%-72s grid=%-9d %-28s n=%-4d mean=%.1f" % (name, grid, c, len(v), sum(v) / len(v)))
