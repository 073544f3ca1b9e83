#!/bin/bash
# GPU box (through gpurun): same-box A/B of two builds of libteal_hip.so with scripts/micro/layer_bench — used in round 5 for the
# "partial last batch requested inside the stream pipeline" variant of teal_gemv_fast.h (rejected: profiles/r05_partial_batch_ab.txt).
# Before the call, in the container: keep the baseline library as teal_amd/libteal_hip_base.so and link a second binary against it
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include scripts/micro/layer_bench.cpp -L teal_amd -l:libteal_hip_base.so -ldl \
#         -Wl,-rpath,'$ORIGIN/../../teal_amd' -o scripts/micro/layer_bench_base
# then build the variant into teal_amd/libteal_hip.so and scripts/micro/layer_bench as usual.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_partial_batch_ab.txt; : > $OUT
for r in 1 2 3; do for b in layer_bench_base layer_bench; do for m in "" "--dense"; do
echo "== $b $m (run $r)" >> $OUT; timeout 300 scripts/micro/$b --model 7b --no_pair --layers 32 --steps 100 $m 2>/dev/null | grep -i "us/layer\|per layer\|tok/s\|token" | head -4 >> $OUT; done; done; done
echo "== verify" >> $OUT; LB_VERIFY=3 timeout 300 scripts/micro/layer_bench --model 7b --no_pair --layers 4 --steps 5 2>&1 | grep -i "verify\|identical\|mismatch\|differ" | head >> $OUT
echo "== 70b" >> $OUT
for b in layer_bench_base layer_bench; do echo "== $b 70b" >> $OUT; timeout 300 scripts/micro/$b --model 70b --layers 8 --steps 40 2>/dev/null | grep -i "us/layer\|per layer\|tok/s" | head -3 >> $OUT; done
cat $OUT
