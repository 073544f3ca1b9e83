cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kind in all; do
  echo "== engine steps next to a process running torch elementwise kernels ($kind)"
  python scripts/micro/_elementwise_noise.py $kind 45 &
  sleep 10
  timeout 200 python scripts/micro/concurrency_determinism_probe.py --noise none --repeats 25000 2>&1 | grep "^\[p" | cut -c1-330 | tail -8
  wait
done
echo "== trivial register kernel next to the same process"
python scripts/micro/_elementwise_noise.py all 30 &
sleep 10
scripts/micro/concurrent_vgpr_probe 40000 0 next-to-elementwise
wait
