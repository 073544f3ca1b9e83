#!/bin/bash
# BASELINE.json configs 1-5 on the GPU box (through gpurun). Logs under gpurun_out/configs/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/configs; mkdir -p $OUT
G="python -m teal_amd.gpt_fast.generate --compile --engine --num_samples 3 --max_new_tokens 200 --report_kept"
# the reference-harness flow (model(token, pos) -> torch sampler under a hipGraph; single-token calls run the fused step)
GD="python -m teal_amd.gpt_fast.generate --compile --no_engine --num_samples 3 --max_new_tokens 200"
echo "== config 1: scripts/benchmark_gemv.py"
python scripts/benchmark_gemv.py --out_size 4096 --cpu --out $OUT > $OUT/c1_gemv_4096x4096.log 2>&1; tail -22 $OUT/c1_gemv_4096x4096.log | grep -E "s=0.00|s=0.25|s=0.50|s=0.75|s=0.90"
python scripts/benchmark_gemv.py --out_size 14336 --cpu --out $OUT > $OUT/c1_gemv_4096x14336.log 2>&1; grep -E "s=0.00|s=0.25|s=0.50|s=0.75|s=0.90" $OUT/c1_gemv_4096x14336.log
echo "== config 2: Llama-2-7B fp16 uniform 50%"
$G --synthetic 7B --sparsity 0.5 > $OUT/c2_7b_s50.log 2>&1; grep -E "Average|kept" $OUT/c2_7b_s50.log
$G --synthetic 7B --sparsity 0.0 > $OUT/c2_7b_dense.log 2>&1; grep -E "Average" $OUT/c2_7b_dense.log
$GD --synthetic 7B --sparsity 0.5 > $OUT/c2_7b_s50_reference_harness_flow.log 2>&1; grep -E "Average" $OUT/c2_7b_s50_reference_harness_flow.log
$GD --no_fused_decode --synthetic 7B --sparsity 0.5 --num_samples 2 > $OUT/c2_7b_s50_op_by_op.log 2>&1; grep -E "Average" $OUT/c2_7b_s50_op_by_op.log
echo "== config 3: Llama-3-8B bf16 uniform 40%"
$G --synthetic llama-3-8b --precision bf16 --sparsity 0.4 > $OUT/c3_8b_bf16_s40.log 2>&1; grep -E "Average|kept" $OUT/c3_8b_bf16_s40.log
$G --synthetic llama-3-8b --precision bf16 --sparsity 0.0 > $OUT/c3_8b_bf16_dense.log 2>&1; grep -E "Average" $OUT/c3_8b_bf16_dense.log
echo "== config 4: Llama-2-7B fp16 block-wise greedy (target 0.5)"
$G --synthetic 7B --sparsity 0.5 --greedy_lookup tests/golden/greedy_llama2_7b.json > $OUT/c4_7b_greedy50.log 2>&1; grep -E "Average|kept" $OUT/c4_7b_greedy50.log
if [ "${SKIP_70B:-0}" != "1" ]; then
echo "== config 5: Llama-2-70B fp16 uniform 50%"
$G --synthetic 70B --sparsity 0.5 --num_samples 2 > $OUT/c5_70b_s50.log 2>&1; grep -E "Average|kept|Memory|Error|error" $OUT/c5_70b_s50.log | head
$G --synthetic 70B --sparsity 0.0 --num_samples 2 > $OUT/c5_70b_dense.log 2>&1; grep -E "Average|Error|error" $OUT/c5_70b_dense.log | head
fi
