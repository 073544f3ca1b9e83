cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in ${VARIANTS:-product wait noprio plain}; do
  echo "== library variant: $v"
  if [ $v = product ]; then unset TEAL_LIB_PATH; else export TEAL_LIB_PATH=$PWD/teal_amd/libteal_hip_exp_$v.so; fi
  timeout 200 python scripts/micro/concurrency_determinism_probe.py --noise ${NOISE:-op_linear_qkv} --repeats ${REPEATS:-1500} 2>&1 | grep "^\[p" | cut -c1-420 | tail -${TAIL:-2}
done
