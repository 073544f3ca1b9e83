#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output (kernel trace + PMC counters) into small per-kernel tables.

Launches are keyed by (kernel instantiation, workgroups): one instantiation of the lean GEMV serves several launches of
a decode step (the unpaired gate|up launch and the lm_head launch differ only in the grid), and a per-instantiation
average would mix them.  Writes kernel_stats_top.csv (from rocprofv3's own --stats table, per instantiation),
kernel_stats_by_grid.csv (per instantiation AND grid, from the kernel trace) and pmc_{fetch,write}_by_kernel.csv.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 24)


def short(n):
    for kn in ("gemv_fast_kernel", "sparse_gemv_kernel", "decode_attention_split_kernel", "decode_attention_gqa_kernel",
               "decode_attention_merge_kernel", "sample_topk_window_kernel", "sample_topk_multi_kernel", "sparse_gemv_int4_kernel"):
        m = re.search(kn + r"<([^>]*)>", n)
        if m:
            return kn + "<" + m.group(1).replace(" ", "") + ">"
    for k in ("decode_attention_kernel", "sample_topk_kernel", "splitk_reduce_kernel", "compact_kernel", "gateup_silu_epilogue_kernel"):
        if k in n:
            return k
    return n[:70]


def main(root):
    out = []
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        out.append("== kernel stats (rocprofv3 --kernel-trace --stats), top 12 by total time")
        out.append("%-66s %8s %12s %10s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows[:12]:
            out.append("%-66s %8s %12.3f %10.2f %8s" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
        with open(os.path.join(root, "kernel_stats_top.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows[:40]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        agg = defaultdict(lambda: [0, 0, 1 << 62, 0])
        for r in csv.DictReader(open(f)):
            wgs = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * (int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg[(short(r["Kernel_Name"]), wgs)]
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        tot = sum(a[1] for a in agg.values()) or 1
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        out.append("== kernel trace by (instantiation, workgroups), top 14 by total time")
        out.append("%-66s %6s %8s %12s %10s %7s" % ("kernel", "wgs", "calls", "total_ms", "avg_us", "pct"))
        for (k, wgs), a in rows[:14]:
            out.append("%-66s %6d %8d %12.3f %10.2f %7.2f" % (k, wgs, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot))
        with open(os.path.join(root, "kernel_stats_by_grid.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "workgroups", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for (k, wgs), a in rows[:40]:
                w.writerow([k, wgs, a[0], a[1], "%.1f" % (a[1] / a[0]), "%.3f" % (100.0 * a[1] / tot), a[2], a[3]])
    for tag in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
                k = (short(r["Kernel_Name"]), wgs, r["Counter_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
            out.append(f"== {tag}: per-dispatch average counter value (raw units as reported), by (instantiation, workgroups)")
            with open(os.path.join(root, f"{tag}_by_kernel.csv"), "w") as g:
                w = csv.writer(g)
                w.writerow(["kernel", "workgroups", "counter", "dispatches", "avg_value"])
                for (k, wgs, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
                    w.writerow([k, wgs, c, n, v / n])
                    out.append("%-66s %6d %-12s n=%6d avg=%14.1f" % (k, wgs, c, n, v / n))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
