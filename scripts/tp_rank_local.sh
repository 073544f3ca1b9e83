#!/bin/bash
# GPU box (through gpurun): ONE rank's launch sequence under tensor parallelism, timed on one GPU with layer_bench --tp W
# (rank-local widths: gpt-fast/tp.py:110-140; the two all-reduces per layer are NOT part of it) next to the unsharded step.
# Output: gpurun_out/r05_tp_rank_local_launches.txt (copied to profiles/ by hand).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_tp_rank_local_launches.txt
: > $OUT
run() { echo "== layer_bench $*" >> $OUT; LB_DESC=1 timeout 300 scripts/micro/layer_bench "$@" >> $OUT 2>/dev/null; echo "rc=$?" >> $OUT; }
# (--no_pair where DecodeEngine runs gate | up unpaired: everything but unsharded 70B-class widths)
run --model 7b --no_pair --layers 16 --steps 60 --phase
run --model 7b --no_pair --tp 2 --layers 16 --steps 60 --phase
run --model 8b --bf16 --no_pair --layers 16 --steps 60 --phase
run --model 8b --bf16 --no_pair --tp 2 --layers 16 --steps 60 --phase
run --model 70b --layers 8 --steps 40 --phase
run --model 70b --no_pair --tp 8 --layers 16 --steps 60 --phase
cut -c1-260 $OUT | grep -v "per-wave\|tail:" | tail -80
