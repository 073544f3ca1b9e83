#!/bin/bash
# Pure-HIP reproducer (none of the product's code): the GEMV-like stream kernel of concurrent_stream_probe.hip with its accumulators
# advanced by v_pk_fma_f32 (mode 5), launch after launch compared with the first launch's bits,
#   alone / next to another process's skinny rocBLAS GEMM (the known trigger) / next to a pure-HIP MFMA aggressor in another process /
#   with that aggressor on a second stream of the same process; mode 0 (the same kernel without packed math) as the control.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=scripts/micro/concurrent_stream_probe
N=${LAUNCHES:-20000}
echo "== alone"
$P 5 $N alone | tail -1
echo "== next to another process running F.linear(x[1,24,4096], W_qkv[12288,4096]) (torch / Tensile stream-K)"
rm -f /tmp/noise_ready
NOISE_READY_FILE=/tmp/noise_ready python scripts/micro/concurrency_determinism_probe.py --noise-child op_linear_qkv > /dev/null 2>&1 &
NP=$!
for i in $(seq 1 200); do [ -f /tmp/noise_ready ] && break; sleep 1; done
$P 5 $N next-to-F.linear | tail -4 | cut -c1-300
$P 0 $N next-to-F.linear-control-not-packed | tail -1
kill $NP; wait $NP 2>/dev/null
echo "== next to a pure-HIP MFMA aggressor in ANOTHER process (one-wave workgroups of v_mfma_f32_16x16x16_f16 loops)"
for cfg in "2048 4000" "8192 500" "512 20000"; do
  $P 6 0 agg $cfg > /dev/null 2>&1 &
  AP=$!
  sleep 3
  $P 5 $N "next-to-mfma($cfg)" | tail -3 | cut -c1-300
  kill $AP; wait $AP 2>/dev/null
done
echo "== the same aggressor on a second stream of the SAME process"
$P 7 $N same-process-two-streams 2048 4000 | tail -3 | cut -c1-300
