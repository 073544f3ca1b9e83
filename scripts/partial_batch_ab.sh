cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_partial_batch_ab.txt; : > $OUT
for r in 1 2 3; do for b in layer_bench_base layer_bench; do for m in "" "--dense"; do
echo "== $b $m (run $r)" >> $OUT; timeout 300 scripts/micro/$b --model 7b --no_pair --layers 32 --steps 100 $m 2>/dev/null | grep -i "us/layer\|per layer\|tok/s\|token" | head -4 >> $OUT; done; done; done
echo "== verify" >> $OUT; LB_VERIFY=3 timeout 300 scripts/micro/layer_bench --model 7b --no_pair --layers 4 --steps 5 2>&1 | grep -i "verify\|identical\|mismatch\|differ" | head >> $OUT
echo "== 70b" >> $OUT
for b in layer_bench_base layer_bench; do echo "== $b 70b" >> $OUT; timeout 300 scripts/micro/$b --model 70b --layers 8 --steps 40 2>/dev/null | grep -i "us/layer\|per layer\|tok/s" | head -3 >> $OUT; done
cat $OUT
