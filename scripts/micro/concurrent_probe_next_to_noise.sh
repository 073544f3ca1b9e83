cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -f /tmp/noise_ready
NOISE_READY_FILE=/tmp/noise_ready python scripts/micro/concurrency_determinism_probe.py --noise-child ${NOISE:-op_linear_qkv} > /dev/null 2>&1 &
NP=$!
for i in $(seq 1 200); do [ -f /tmp/noise_ready ] && break; sleep 1; done
echo "noise ${NOISE:-op_linear_qkv} ready after $i s"
for mode in ${MODES:-0 3 4}; do
  timeout 120 scripts/micro/concurrent_stream_probe $mode ${LAUNCHES:-40000} next-to-noise | cut -c1-400 | tail -6
done
echo "== the engine probe next to the same noise process (control: the trigger is active)"
timeout 100 python scripts/micro/concurrency_determinism_probe.py --noise none --repeats 2000 2>&1 | grep "^\[p" | cut -c1-200 | tail -2
kill $NP
