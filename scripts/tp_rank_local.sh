#!/bin/bash
# GPU box (through gpurun): ONE rank's launch sequence under tensor parallelism, timed on one GPU with layer_bench --tp W
# (rank-local widths: gpt-fast/tp.py:110-140; the two all-reduces per layer are NOT part of it) next to the unsharded step.
# Output: gpurun_out/r06_tp_rank_local_launches.txt (copied to profiles/ by hand).  Round 6 adds the --presum runs: wo / down fold
# their row slices themselves (TEAL_OUT_SLAB_SUM, arrival tickets) and hand over ONE fp32 [dim] vector — the all-reduce payload a
# rank would send drops from [dim][4..8] fp32 (64-256 KB) to [dim] fp32 (16-32 KB); the price is the launch's ticket round.
# layer_bench links against the diagnostics build (LB_DESC, --phase):
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include scripts/micro/layer_bench.cpp -DTEAL_DIAGNOSTICS -L teal_amd -lteal_hip_diag \
#         -ldl -Wl,-rpath,'$ORIGIN/../../teal_amd' -o scripts/micro/layer_bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_tp_rank_local_launches.txt
: > $OUT
run() { echo "== layer_bench $*" >> $OUT; LB_DESC=1 timeout 300 scripts/micro/layer_bench "$@" >> $OUT 2>/dev/null; echo "rc=$?" >> $OUT; }
# (--no_pair where DecodeEngine runs gate | up unpaired: everything but unsharded 70B-class widths)
run --model 7b --no_pair --layers 16 --steps 60 --phase
run --model 7b --no_pair --tp 2 --layers 16 --steps 60 --phase
run --model 7b --no_pair --presum --layers 16 --steps 60
run --model 7b --no_pair --tp 2 --presum --layers 16 --steps 60
run --model 8b --bf16 --no_pair --layers 16 --steps 60 --phase
run --model 8b --bf16 --no_pair --tp 2 --layers 16 --steps 60 --phase
run --model 70b --layers 8 --steps 40 --phase
run --model 70b --no_pair --tp 8 --layers 16 --steps 60 --phase
run --model 70b --presum --layers 8 --steps 40
run --model 70b --no_pair --tp 8 --presum --layers 16 --steps 60
cut -c1-260 $OUT | grep -v "per-wave\|tail:" | tail -80
