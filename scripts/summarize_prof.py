#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output (kernel stats + PMC counters) into small per-kernel tables."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    for kn in ("gemv_fast_kernel", "sparse_gemv_kernel", "decode_attention_split_kernel", "decode_attention_gqa_kernel",
               "decode_attention_merge_kernel", "sample_topk_window_kernel", "sample_topk_multi_kernel"):
        m = re.search(kn + r"<([^>]*)>", n)
        if m:
            return kn + "<" + m.group(1).replace(" ", "") + ">"
    for k in ("decode_attention_kernel", "sparse_gemv_int4_kernel", "sample_topk_kernel", "splitk_reduce_kernel", "compact_kernel", "gateup_silu_epilogue_kernel"):
        if k in n:
            return k
    return n[:70]


def main(root):
    out = []
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        out.append("== kernel stats (rocprofv3 --kernel-trace --stats), top 12 by total time")
        out.append("%-62s %8s %12s %10s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows[:12]:
            out.append("%-62s %8s %12.3f %10.2f %8s" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
        with open(os.path.join(root, "kernel_stats_top.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows[:40]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    for tag in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(lambda: [0, 0.0])
            rd = csv.DictReader(open(f))
            for r in rd:
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
            out.append(f"== {tag}: per-dispatch average counter value (raw units as reported)")
            with open(os.path.join(root, f"{tag}_by_kernel.csv"), "w") as g:
                w = csv.writer(g)
                w.writerow(["kernel", "counter", "dispatches", "avg_value"])
                for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
                    w.writerow([k, c, n, v / n])
                    out.append("%-62s %-12s n=%6d avg=%14.1f" % (k, c, n, v / n))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
