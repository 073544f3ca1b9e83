#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes over the HEADLINE decode step of bench.py — no dense leg, no context
# sweep, no CPU baseline; bench.py --profile-markers brackets the timed hipGraph replays with marker dispatches and
# scripts/summarize_prof.py drops everything outside them (threshold calibration, warm-up, the roofline graph).
# Raw output under gpurun_out/, the summaries are copied into profiles/ afterwards.
#   pass 1: --kernel-trace --stats            (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE                  (HBM read bytes; own pass, kernel-trace only)
#   pass 3: --pmc WRITE_SIZE
# PASSES=trace runs pass 1 only; PASS_TIMEOUT bounds each pass (seconds); BENCH_ARGS e.g. "--model 70B".
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_${ROUND:-r04}${TAG:-}
rm -rf "$OUT"; mkdir -p "$OUT"
COMMON="--no-dense --no-cpu-baseline --no-context-sweep --no-live-traffic --profile-markers ${BENCH_ARGS:-}"
T="timeout ${PASS_TIMEOUT:-240}"   # every pass bounded: a pass that stalls must not eat the box's time limit
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python bench.py --steps ${STEPS:-100} --warmup 10 $COMMON > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
if [ "${PASSES:-all}" = "all" ]; then
# Counter passes run with AMD_SERIALIZE_KERNEL=3 (the HIP runtime waits before and after every launch; the counters of a launch
# do not depend on it).  Without it rocprofv3 7.2.0's counter mode — which serialises dispatches itself and adds its own packets
# around each — falls behind this command's asynchronous launch stream (thousands of dispatches between synchronisations), the
# intercepted queue fills, and the process dies in librocprofiler-sdk's queue write interceptor reading one packet past the end
# of the 1 MiB ring (SIGSEGV) or with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT: 9 of 10 unserialised passes of the round's last
# builds ended that way (gdb backtrace in profiles/README.md), every serialised one completed.
for C in FETCH_SIZE WRITE_SIZE; do
  tag=pmc_$(echo ${C%_SIZE} | tr A-Z a-z)
  AMD_SERIALIZE_KERNEL=3 timeout ${PMC_TIMEOUT:-120} rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$tag" -o bench -- python bench.py --steps 10 --warmup 2 $COMMON > "$OUT/$tag.log" 2>&1
  echo "$tag rc=$?"
done
fi
find "$OUT" -name "*.db" -delete
python scripts/summarize_prof.py "$OUT" "${MODEL:-7B}" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
grep '^{"metric"' "$OUT/trace.log" | tail -1 > "$OUT/bench_line_under_trace.json"
# keep the merge-back small: drop the big traces, keep stats + counters aggregated by the summary
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
find "$OUT" -name "*agent_info.csv" -delete
