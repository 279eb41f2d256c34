#!/usr/bin/env python3
"""Condenses a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a short markdown
table (kernel names truncated) for profiles/.  Usage: summarize_rocprof.py <dir> [title]"""
import csv
import glob
import os
import re
import sys


def short(name, n=90):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= n else name[:n] + "..."


def main():
    d = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else d
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    assert files, "no *kernel_stats.csv under " + d
    print("# rocprofv3 kernel stats: %s\n" % title)
    for f in files:
        rows = list(csv.DictReader(open(f)))
        print("| kernel | calls | avg us | min us | max us | total ms | % |")
        print("|---|---|---|---|---|---|---|")
        for r in rows[:25]:
            print("| `%s` | %s | %.2f | %.2f | %.2f | %.3f | %s |" % (
                short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        print()


if __name__ == "__main__":
    main()
