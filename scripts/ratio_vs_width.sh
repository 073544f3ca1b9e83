#!/bin/bash
# Where does the path meet the north-star targets (>= 1.8x over dense, >= 0.70 of 8 TB/s on the dominant launch)?
# bench.py on Llama-2 7B / 13B / 30B / 34B / 70B at 50 % and on 7B at 40 / 60 / 70 % (same engine, same dense comparator:
# every row kept), each line with its floor_model (fixed vs streaming microseconds per layer).  Run through gpurun:
#   gpurun --timeout 2400 -- 'bash scripts/ratio_vs_width.sh'
# Lines under gpurun_out/ratio/; scripts/ratio_table.py turns them into profiles/r06_ratio_vs_width.txt.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/ratio; mkdir -p $OUT
COMMON="--no-cpu-baseline --no-context-sweep --no-live-traffic --no-reference-dense --steps ${STEPS:-100} --warmup 10"
run() {  # tag, args...
  local tag=$1; shift
  timeout ${RUN_TIMEOUT:-900} python bench.py $COMMON "$@" > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
  grep '^{"metric"' $OUT/$tag.log | tail -1 > $OUT/$tag.json
}
for s in ${SPARSITIES-0.4 0.5 0.6 0.7}; do run 7B_s$s --model 7B --sparsity $s; done
for m in ${MODELS-13B 30B 34B 70B}; do run ${m}_s0.5 --model $m --sparsity 0.5; done
python scripts/ratio_table.py $OUT | tee $OUT/table.txt
