#!/usr/bin/env python3
"""Condenses a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a short markdown
table (kernel names truncated) for profiles/.  Usage: summarize_rocprof.py <dir> [title] [kernel-substring]
With a kernel substring (and the *kernel_trace.csv still present) it also reports that kernel's duration over its longest
run of back-to-back launches -- for bench.py that is the replayed tape incl. the timed region, the launches the HIP-event
figure of the bench line is taken over; the table's average is over ALL launches of the run (recording pass and actor loop
included, where the policy's kernels run in between)."""
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
    if len(sys.argv) > 3:
        traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if traces:
            rows = sorted(csv.DictReader(open(traces[0])), key=lambda r: int(r["Start_Timestamp"]))
            best, cur = [], []
            for r in rows:
                if sys.argv[3] in r["Kernel_Name"]:
                    cur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                else:
                    best, cur = (cur if len(cur) > len(best) else best), []
            best = cur if len(cur) > len(best) else best
            if best:
                print("`%s`, longest run of back-to-back launches: %d launches, avg %.2f us, min %.2f, max %.2f\n" % (
                    sys.argv[3], len(best), sum(best) / len(best) / 1e3, min(best) / 1e3, max(best) / 1e3))


if __name__ == "__main__":
    main()
