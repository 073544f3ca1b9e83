#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output (kernel trace + PMC counters) of `bench.py --profile-markers` into small per-kernel tables.

* Only the TIMED hipGraph replays count: bench.py brackets the timed region with one marker dispatch each (compact_kernel, a
  name nothing in a decode step uses); dispatches outside the two markers — threshold calibration, eager warm-up, the
  roofline graph — are dropped (kernel trace: by timestamp; counter passes: by dispatch id).  Without markers everything
  counts (and the summary says so).
* Launches are keyed by (kernel instantiation, workgroups): one instantiation of the lean GEMV serves several launches of
  a decode step (the unpaired gate|up launch and the lm_head launch differ only in the grid).
* The roofline table prices every launch of the decode step against SURVEY 8(d)'s algorithmic bytes at the kept fractions
  bench.py measured in the same run (its JSON line, trace.log): nnz * N * 2 + Z * 2 + N * 2 per GEMV.

Writes kernel_stats_by_grid.csv, pmc_{fetch,write}_by_kernel.csv, roofline_by_launch.csv next to the raw output and prints
the summary (profiles/rNN_bench_summary.txt).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 24)
MARKER = "compact_kernel"
HBM_PEAK = 8.0e12


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


def bench_line(root):
    """the JSON line bench.py printed under the trace pass (kept fractions, config)"""
    for f in ("trace.log",):
        p = os.path.join(root, f)
        if os.path.exists(p):
            for line in reversed(open(p, errors="replace").read().splitlines()):
                if line.startswith("{") and '"metric"' in line:
                    try:
                        return json.loads(line)
                    except json.JSONDecodeError:
                        pass
    return None


SHAPES = {"7B": (4096, 11008, 32, 32, 128, 32000), "llama-3-8b": (4096, 14336, 32, 8, 128, 128256), "70B": (8192, 28672, 64, 8, 128, 32000),
          "13B": (5120, 13824, 40, 40, 128, 32000)}


def launch_table(line, model):
    """(label, predicate(kernel, wgs), algorithmic bytes) of every launch kind of the decode step"""
    dim, inter, nh, nkv, hd, vocab = SHAPES[model]
    kv = nkv * hd
    nqkv = dim + 2 * kv
    kf = line.get("kept_fraction", {}) if line else {}
    g = lambda p: float(kf.get(p, 1.0))  # noqa: E731
    pos = 0.0
    if line:
        m = re.match(r"(\d+)\.\.(\d+)", str(line.get("config", {}).get("context_positions", "")))
        if m:
            pos = (int(m.group(1)) + line.get("warmup", 0) + int(m.group(2))) / 2.0  # middle of the timed range
    mode_of = lambda k: (re.search(r"gemv_fast_kernel<(?:true|false),(\d)", k) or [None, None])[1]  # noqa: E731
    return [
        ("gate|up", lambda k, w: mode_of(k) == "1" and w in (2 * inter // 128, inter // 64, inter // 128),
         (g("gate") + g("up")) * dim * inter * 2 + dim * 2 + 2 * inter * 2),
        ("qkv", lambda k, w: mode_of(k) == "1" and w in (nqkv // 64, 2 * (nqkv // 64), nqkv // 128 * 3, nqkv // 128 * 2),
         (g("q") * dim + g("k") * kv + g("v") * kv) * dim * 2 + dim * 2 + nqkv * 2),
        ("down", lambda k, w: mode_of(k) in ("2", "3"), g("down") * inter * dim * 2 + inter * 2 + dim * 2),
        ("wo", lambda k, w: mode_of(k) in ("4", "0"), g("o") * dim * dim * 2 + dim * 2 + dim * 2),
        ("attention", lambda k, w: k.startswith("decode_attention_split_kernel") or k.startswith("decode_attention_gqa_kernel"),
         2.0 * nkv * (pos + 1) * hd * 2 + dim * 2 * 2),
        # (a 128 k-entry vocabulary takes 512-column tiles: the general kernel, sparse_gemv_kernel<64,...>, ceil(vocab / 512) workgroups)
        ("lm_head", lambda k, w: (mode_of(k) == "1" and w in (vocab // 128, vocab // 64, vocab // 256)) or
         (k.startswith("sparse_gemv_kernel<64,") and w == -(-vocab // 512)), dim * vocab * 2 + dim * 2 + vocab * 2),
        ("sampler", lambda k, w: k.startswith("sample_topk"), vocab * 2),
    ]


def main(root, model="7B"):
    out = []
    line = bench_line(root)
    window, has_markers = None, False
    trace_rows = []
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        trace_rows += list(csv.DictReader(open(f)))
    marks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in trace_rows if MARKER in r["Kernel_Name"])
    if len(marks) >= 2:
        window, has_markers = (marks[0][1], marks[-1][0]), True
    out.append("== scope: " + (f"the {line['steps']} timed hipGraph replays between bench.py's two marker dispatches "
                               f"({(window[1] - window[0]) / 1e3 / line['steps']:.1f} us per replay incl. host gaps)" if has_markers and line
                               else "EVERY dispatch of the process (no marker dispatches found)"))
    agg = defaultdict(lambda: [0, 0, 1 << 62, 0])
    for r in trace_rows:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if window and not (window[0] <= st and en <= window[1]):
            continue
        wgs = 1
        for ax in "XYZ":  # every axis: the split attention kernel's grid is (split, query head of the group, KV head)
            wgs *= int(r.get("Grid_Size_" + ax, 1) or 1) // max(1, int(r.get("Workgroup_Size_" + ax, 1) or 1))
        d = en - st
        a = agg[(short(r["Kernel_Name"]), wgs)]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if rows:
        out.append("== kernel trace by (instantiation, workgroups), by total time")
        out.append("%-72s %6s %8s %12s %10s %7s" % ("kernel", "wgs", "calls", "total_ms", "avg_us", "pct"))
        for (k, wgs), a in rows[:14]:
            out.append("%-72s %6d %8d %12.3f %10.2f %7.2f" % (k, wgs, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot))
        with open(os.path.join(root, "kernel_stats_by_grid.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "workgroups", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for (k, wgs), a in rows[:40]:
                w.writerow([k, wgs, a[0], a[1], "%.1f" % (a[1] / a[0]), "%.3f" % (100.0 * a[1] / tot), a[2], a[3]])
    pmc = {}
    for tag in ("pmc_fetch", "pmc_write"):
        crow = []
        for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
            crow += list(csv.DictReader(open(f)))
        if not crow:
            continue
        mids = sorted(int(r["Dispatch_Id"]) for r in crow if MARKER in r["Kernel_Name"])
        cwin = (mids[0], mids[-1]) if len(mids) >= 2 else None
        cagg = defaultdict(lambda: [0, 0.0])
        for r in crow:
            if cwin and not (cwin[0] < int(r["Dispatch_Id"]) < cwin[1]):
                continue
            wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            k = (short(r["Kernel_Name"]), wgs, r["Counter_Name"])
            cagg[k][0] += 1
            cagg[k][1] += float(r["Counter_Value"])
        out.append(f"== {tag}: per-dispatch average counter value (KiB as reported), by (instantiation, workgroups)"
                   + ("" if cwin else "  [no markers: every dispatch]"))
        with open(os.path.join(root, f"{tag}_by_kernel.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["kernel", "workgroups", "counter", "dispatches", "avg_value"])
            for (k, wgs, c), (n, v) in sorted(cagg.items(), key=lambda kv: -kv[1][1])[:30]:
                w.writerow([k, wgs, c, n, v / n])
                if len(out) < 200 and n >= 8:
                    out.append("%-72s %6d %-12s n=%6d avg=%14.1f" % (k, wgs, c, n, v / n))
                pmc[(tag, k, wgs)] = v / n
    if line and rows and model in SHAPES:
        out.append("== roofline by launch of the decode step (SURVEY 8(d) bytes at the kept fractions of this run: "
                   + ", ".join(f"{k} {v:.3f}" for k, v in line.get("kept_fraction", {}).items()) + ")")
        out.append("%-10s %-62s %5s %9s %9s %8s %10s %10s" % ("launch", "kernel", "wgs", "algo MB", "avg us", "of 8TB/s", "FETCHx2 MB", "/ algo"))
        with open(os.path.join(root, "roofline_by_launch.csv"), "w") as g:
            w = csv.writer(g)
            w.writerow(["launch", "kernel", "workgroups", "calls", "algorithmic_bytes", "avg_us", "frac_of_8TBps", "fetch_x2_bytes", "fetch_over_algorithmic"])
            per_token = 0.0
            for label, pred, nbytes in launch_table(line, model):
                hit = [((k, wgs), a) for (k, wgs), a in rows if pred(k, wgs)]
                if not hit:
                    continue
                (k, wgs), a = hit[0]
                us = a[1] / a[0] / 1e3
                per_token += a[1] / max(1, line["steps"]) / 1e3
                fetch = pmc.get(("pmc_fetch", k, wgs))
                fb = fetch * 1024 * 2 if fetch is not None else None
                w.writerow([label, k, wgs, a[0], int(nbytes), "%.2f" % us, "%.3f" % (nbytes / (us * 1e-6) / HBM_PEAK),
                            "" if fb is None else int(fb), "" if fb is None else "%.3f" % (fb / nbytes)])
                out.append("%-10s %-62s %5d %9.2f %9.2f %8.3f %10s %10s" % (label, k[:62], wgs, nbytes / 1e6, us, nbytes / (us * 1e-6) / HBM_PEAK,
                                                                          "-" if fb is None else "%.2f" % (fb / 1e6), "-" if fb is None else "%.3f" % (fb / nbytes)))
            out.append(f"   sum of these launches per token: {per_token:.1f} us of kernel time; bench line: {line['ms_per_step'] * 1e3:.1f} us per step "
                       f"= {line['value']:.1f} tokens/s")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "7B")
