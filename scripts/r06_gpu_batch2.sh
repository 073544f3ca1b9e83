#!/bin/bash
# round 6, GPU call 2: rocprofv3 evidence for configs 3 and 5 at the headline's standard (marker-bracketed kernel trace, FETCH_SIZE
# and WRITE_SIZE passes), one rank's TP launches incl. the presummed hand-over, the prompt pass by prompt length
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== profile 8B bf16 @ 40 %"
ROUND=r06 TAG=_8b MODEL=llama-3-8b BENCH_ARGS="--model llama-3-8b --precision bf16 --sparsity 0.4" PASS_TIMEOUT=420 PMC_TIMEOUT=300 bash scripts/profile_bench.sh > gpurun_out/r06_profile_8b.log 2>&1
tail -25 gpurun_out/r06_profile_8b.log
echo "== profile 70B fp16 @ 50 %"
ROUND=r06 TAG=_70b MODEL=70B BENCH_ARGS="--model 70B --sparsity 0.5" STEPS=60 PASS_TIMEOUT=900 PMC_TIMEOUT=900 bash scripts/profile_bench.sh > gpurun_out/r06_profile_70b.log 2>&1
tail -25 gpurun_out/r06_profile_70b.log
echo "== tp rank-local launches (+ presummed hand-over)"
bash scripts/tp_rank_local.sh > /dev/null 2>&1; grep -c "" gpurun_out/r06_tp_rank_local_launches.txt
echo "== prompt pass by prompt length"
timeout 900 python scripts/prefill_vs_prompt_length.py > gpurun_out/r06_prefill_vs_prompt_length.txt 2>&1; cat gpurun_out/r06_prefill_vs_prompt_length.txt | tail -14
echo "== nccl probe (both capture modes)"; timeout 400 python scripts/micro/nccl_one_rank_probe.py > gpurun_out/r06_nccl_one_rank_probe.txt 2>&1; echo "rc=$?"; grep -v "^frame\|^$" gpurun_out/r06_nccl_one_rank_probe.txt | cut -c1-200 | tail -30
echo "== rccl one-rank tests + TP tests"; timeout 1200 python -m pytest tests/test_rccl_one_rank.py tests/test_tp_gpu.py tests/test_harness.py -m gpu -q -x -s --timeout 900 > gpurun_out/r06_rccl_tp_tests.txt 2>&1; echo "rc=$?"; grep '^{"backend"' gpurun_out/r06_rccl_tp_tests.txt | cut -c1-1500; tail -5 gpurun_out/r06_rccl_tp_tests.txt
echo "== GPU suite against the diagnostics build"; TEAL_LIB_FLAVOR=diag timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r06_gpu_tests_diag_build.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/r06_gpu_tests_diag_build.txt
