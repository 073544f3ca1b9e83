#!/bin/bash
# The levers' code exists only in commit 8544824 (both measured slower: profiles/r05_layer_experiments.txt); to re-run, check that
# commit out and build:  python -c "from teal_amd import _lib; _lib.build(out='teal_amd/libteal_hip_r05exp.so', extra_flags=('-DTEAL_R05_EXPERIMENTS',))"
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 -DTEAL_R05_EXPERIMENTS -I include scripts/micro/layer_bench.cpp -L teal_amd \
#         -l:libteal_hip_r05exp.so -ldl -Wl,-rpath,'$ORIGIN/../../teal_amd' -o scripts/micro/layer_bench_r05exp
# GPU box (through gpurun): the two round-5 T1 levers, in-process A/B of the whole token through scripts/micro/layer_bench_r05exp
# (built with -DTEAL_R05_EXPERIMENTS against teal_amd/libteal_hip_r05exp.so; the product library has neither switch nor code):
#   (a) LB_FOLDAB=1  the attention of a head inside the qkv launch (K / V prefetched at kernel entry, arrival counters per head)
#   (b) LB_SLIMAB=1  wo / down add the residual themselves (tickets), consumers read h;  =2: plus per-tile sums of h^2 (no barrier)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_levers.txt
: > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout 240 scripts/micro/layer_bench_r05exp --model 7b --no_pair --layers 32 --steps 100 >> $OUT 2>&1; echo "rc=$?" >> $OUT; }
run LB_FOLDAB=1 LB_VERIFY_SOFT=1
run LB_SLIMAB=1
run LB_SLIMAB=2
echo "== dense (every row kept)" >> $OUT
for v in LB_FOLDAB=1 LB_SLIMAB=2; do echo "== $v --dense" >> $OUT; env $v timeout 240 scripts/micro/layer_bench_r05exp --model 7b --no_pair --layers 32 --steps 60 --dense >> $OUT 2>&1; echo "rc=$?" >> $OUT; done
grep -v "^\[mark\]\|A/B round" $OUT
