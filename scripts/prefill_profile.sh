#!/bin/bash
# GPU box (through gpurun): per-kernel durations of the hand-fused prompt pass (6-token prompt, Llama-2-7B fp16) under rocprofv3.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_prefill; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/pf_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from teal_amd.gpt_fast import generate as G
from teal_amd.gpt_fast.prefill import FusedPrefill
m = G.build_synthetic_model(sys.argv[1] if len(sys.argv) > 1 else "7B", "cuda", torch.float16)
G.apply_sparsity(m, sparsity=0.0, hist_path=None, greedy_lookup=None, synthetic=True, decode_calibration=False)
m.max_seq_length = -1; m.setup_caches(1, 64); G.relayout_for_engine(m)
p = torch.randint(0, 32000, (6,), device="cuda", dtype=torch.int)
pre = FusedPrefill(m, graph=False)
for _ in range(6): pre(p)
torch.cuda.synchronize()
print("used", pre.used)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pf -- python /tmp/pf_run.py > $OUT/log.txt 2>&1
echo "rc=$?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "prefill" in r["Name"] or "gemv" in r["Name"]:
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +4M -delete
