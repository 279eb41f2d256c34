#!/usr/bin/env python3
"""Per-category GPU time of ONE steady-state iteration out of a rocprofv3 kernel trace (csv).
Usage: trace_breakdown.py <dir> <marker-kernel-substring> [top]
The iteration is the span between the last two launches of the marker kernel."""
import csv
import glob
import re
import sys
from collections import defaultdict


def category(n):
    if "igemm" in n or "SubTensorOp" in n or "naive_conv" in n or "kernel_grouped_conv" in n or "batched_gemm_xdlops_bwd_weight" in n:
        return "miopen conv"
    for k in ("bias_res_relu", "encoder_fwd_kernel", "encoder_bwd_kernel", "encoder_wgrad_kernel", "conv0_wgrad_kernel", "encoder_pack",
              "env_step_kernel", "comm_mask_kernel", "gather_kernel", "tree_", "reset_kernel", "navi_bfs", "recurrent_wide_bwd_kernel",
              "recurrent_wide_kernel", "recurrent_bwd_kernel", "recurrent_infer_kernel", "flush_"):
        if k in n:
            return "hip: " + k
    if "Cijk" in n:
        return "gemm " + "_".join(n.split("_")[1:3])
    if "gru_cell" in n:
        return "gru pointwise"
    if "softmax" in n:
        return "softmax"
    if "reduce_kernel" in n:
        return "reduce"
    if "elementwise" in n or "vectorized" in n:
        m = re.findall(r"at::native::([A-Za-z_0-9]+)", n)
        return "ew " + (m[1] if len(m) > 1 else m[0] if m else n[:30])
    return "other " + n[:50]


def main():
    d, marker = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    sel = rows[idx[-2]:idx[-1]]
    wall = (int(rows[idx[-1]]["Start_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
    cat = defaultdict(lambda: [0, 0])
    for r in sel:
        c = cat[category(r["Kernel_Name"])]
        c[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c[1] += 1
    busy = sum(v[0] for v in cat.values()) / 1e6
    print("one iteration between launches of `%s`: wall %.2f ms, GPU busy %.2f ms, %d launches\n" % (marker, wall, busy, len(sel)))
    print("| category | ms | launches |\n|---|---|---|")
    for k, v in sorted(cat.items(), key=lambda kv: -kv[1][0])[:top]:
        print("| %s | %.3f | %d |" % (k, v[0] / 1e6, v[1]))
    if len(sys.argv) > 4:  # the largest single launches outside the named hand-written kernels, in launch order
        n = int(sys.argv[4])
        big = sorted(range(len(sel)), key=lambda i: int(sel[i]["Start_Timestamp"]) - int(sel[i]["End_Timestamp"]))[:n]
        print("\n| # in iteration | us | grid | kernel |\n|---|---|---|---|")
        for i in sorted(big):
            r = sel[i]
            print("| %d | %.1f | %s | %s |" % (i, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")),
                                             r["Kernel_Name"][:110]))


if __name__ == "__main__":
    main()
