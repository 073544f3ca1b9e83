#!/usr/bin/env python3
"""gpurun_out/ratio/*.json (bench lines written by scripts/ratio_vs_width.sh) -> the crossover table of
profiles/r06_ratio_vs_width.txt: speed-up over the dense engine, dominant-launch fraction of 8 TB/s, and floor_model's
fixed / streaming split per layer, by model width and by sparsity."""
import glob
import json
import os
import sys


def main(root):
    rows = []
    for f in sorted(glob.glob(os.path.join(root, "*.json"))):
        txt = open(f).read().strip()
        if not txt:
            print(f"# {os.path.basename(f)}: no bench line (see the .log next to it)")
            continue
        d = json.loads(txt)
        fm, rl = d.get("floor_model") or {}, d.get("roofline") or {}
        rows.append((os.path.basename(f)[:-5], d, fm, rl))
    print("%-12s %9s %9s %7s | %-28s %8s %6s | %8s %9s %9s %7s %7s" % (
        "run", "tok/s", "dense", "ratio", "dominant launch (gate|up)", "us", "frac", "fixed us", "stream sp", "stream de", "model", "no-fix"))
    for tag, d, fm, rl in rows:
        lr = fm.get("layer_ratio", {})
        st = fm.get("streaming_us_per_layer", {})
        print("%-12s %9.1f %9.1f %7.3f | %-28s %8.2f %6.3f | %8s %9s %9s %7s %7s" % (
            tag, d["value"], d.get("dense_tokens_per_sec", float("nan")), d.get("speedup_vs_dense", float("nan")),
            "%.1f MB, kept %.3f" % (rl.get("algorithmic_bytes", 0) / 1e6, rl.get("kept_fraction", float("nan"))),
            rl.get("us_per_launch", float("nan")), rl.get("frac", float("nan")),
            fm.get("fixed_us_per_layer", "-"), st.get("sparse", "-"), st.get("dense", "-"),
            lr.get("model (fixed + attention + streaming)", "-"), lr.get("if the fixed per-launch cost and the attention launch were free", "-")))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ratio")
