cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "=== GPU suite (product + diagnostics builds as the tests choose)"
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r06_gpu_suite_c.txt
echo "=== engine next to the aggressors (another process's skinny Tensile GEMMs / 24-token prefill)"
KINDS="op_linear_qkv op_linear_d prefill" REPEATS=3000 bash scripts/concurrency_noise.sh 2>&1 | grep -v "^\[p0\] repeat" | cut -c1-300 | tee gpurun_out/r06_noise_final.txt
echo "=== bench"
python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r06_bench_final_7b.json
python bench.py --model llama-3-8b --precision bf16 --sparsity 0.4 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_bench_final_8b.json
python bench.py --model 70B --steps 60 --warmup 10 --no-cpu-baseline --no-context-sweep --no-reference-dense 2>/dev/null | tail -1 > gpurun_out/r06_bench_final_70b.json
for m in 7b 8b 70b; do python -c "
import json,sys; d=json.load(open('gpurun_out/r06_bench_final_$m.json')); print('$m', round(d['value'],1), d.get('speedup_vs_dense'), d['roofline']['frac'], d['ms_per_step'])"; done
