#!/bin/bash
# What the driver runs at round end, on the final tree: the GPU suite (product build), smoke(), the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "=== GPU suite"
python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r06_gpu_suite_final.txt
echo "=== smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
echo "=== bench (driver's flags)"
( time python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_bench_driver_flags.json ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_driver_flags.json')); print(round(d['value'],1), d.get('speedup_vs_dense'), d['roofline']['frac'], d['ms_per_step'], d['cpu_baseline']['value'], sorted(d.keys()))"
