#!/bin/bash
# round 6, GPU call 1: RCCL one-rank probe, the whole GPU suite on the two-library build, the ratio-vs-width sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== nccl probe"; timeout 300 python scripts/micro/nccl_one_rank_probe.py > gpurun_out/r06_nccl_one_rank_probe.txt 2>&1; echo "rc=$?"; tail -12 gpurun_out/r06_nccl_one_rank_probe.txt
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r06_gpu_tests_batch1.txt 2>&1; echo "rc=$?"; tail -15 gpurun_out/r06_gpu_tests_batch1.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== ratio sweep"; MODELS="${MODELS:-13B 30B 34B 70B}" bash scripts/ratio_vs_width.sh
