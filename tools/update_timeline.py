#!/usr/bin/env python3
"""Timeline of ONE steady-state learner update out of a rocprofv3 kernel trace: every kernel between the last two launches of the
marker kernel (default adam_kernel) in start order -- start offset, duration, gap to the previous kernel's end on the same queue,
queue -- and the span's busy time per queue.  Usage: update_timeline.py <trace dir> [marker] [max lines]"""
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"<.*", "", n)
    return n[-70:]


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    sel = rows[idx[-2] + 1:idx[-1] + 1]
    t0 = int(rows[idx[-2]]["End_Timestamp"])
    last_end = {}
    busy = {}
    print("span %.1f us, %d kernels" % ((int(sel[-1]["End_Timestamp"]) - t0) / 1e3, len(sel)))
    print("| start us | dur us | gap us | queue | kernel |\n|---|---|---|---|---|")
    for r in sel[:top]:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = e
        busy[q] = busy.get(q, 0) + (e - s)
        print("| %8.1f | %7.1f | %6.1f | %s | %s |" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, short(r["Kernel_Name"])))
    print("busy per queue (us):", {q: round(v / 1e3, 1) for q, v in busy.items()})


if __name__ == "__main__":
    main()
