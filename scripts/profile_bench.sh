#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes over bench.py; raw output under gpurun_out/,
# summaries are copied into profiles/ by scripts/summarize_prof.py afterwards.
#   pass 1: --kernel-trace --stats            (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE                  (HBM read bytes; own pass, kernel-trace only)
#   pass 3: --pmc WRITE_SIZE
# PASSES=trace runs pass 1 only; PASS_TIMEOUT bounds each pass (seconds).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_${ROUND:-r03}${TAG:-}
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 40 --warmup 5 --no-dense --no-cpu-baseline ${BENCH_ARGS:-}"
T="timeout ${PASS_TIMEOUT:-200}"   # every pass bounded: a pass that stalls must not eat the box's time limit
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python bench.py $ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
if [ "${PASSES:-all}" = "all" ]; then
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python bench.py --steps 10 --warmup 2 --no-dense --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python bench.py --steps 10 --warmup 2 --no-dense --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
fi
find "$OUT" -name "*.db" -delete
ls -la "$OUT"/*
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merge-back small: drop the big traces, keep stats + counters aggregated by the summary
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
