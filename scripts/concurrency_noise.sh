cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kind in ${KINDS:-randn prefill quantile sdpa cat matmul}; do
  echo "== engine steps next to a process doing: $kind"
  timeout 200 python scripts/micro/concurrency_determinism_probe.py --noise $kind --repeats ${REPEATS:-15000} 2>&1 | grep "^\[p" | cut -c1-330 | tail -5
done
