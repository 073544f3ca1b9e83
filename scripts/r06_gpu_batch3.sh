#!/bin/bash
# round 6, final build (no packed fp32, separate fp16 roundings): rocprofv3 evidence at the headline's standard for the headline
# (Llama-2-7B fp16 @ 50 %), Llama-3-8B bf16 @ 40 % and Llama-2-70B fp16 @ 50 %, and the GPU suite against the diagnostics build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== profile 7B fp16 @ 50 % (headline)"
ROUND=r06 TAG=_7b MODEL=7B BENCH_ARGS="" PASS_TIMEOUT=420 PMC_TIMEOUT=300 bash scripts/profile_bench.sh > gpurun_out/r06_profile_7b.log 2>&1
tail -12 gpurun_out/r06_profile_7b.log
echo "== profile 8B bf16 @ 40 %"
ROUND=r06 TAG=_8b MODEL=llama-3-8b BENCH_ARGS="--model llama-3-8b --precision bf16 --sparsity 0.4" PASS_TIMEOUT=420 PMC_TIMEOUT=300 bash scripts/profile_bench.sh > gpurun_out/r06_profile_8b.log 2>&1
tail -12 gpurun_out/r06_profile_8b.log
echo "== profile 70B fp16 @ 50 %"
ROUND=r06 TAG=_70b MODEL=70B BENCH_ARGS="--model 70B --sparsity 0.5" STEPS=60 PASS_TIMEOUT=900 PMC_TIMEOUT=900 bash scripts/profile_bench.sh > gpurun_out/r06_profile_70b.log 2>&1
tail -12 gpurun_out/r06_profile_70b.log
echo "== GPU suite against the diagnostics build"; TEAL_LIB_FLAVOR=diag timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r06_gpu_tests_diag_build.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/r06_gpu_tests_diag_build.txt
