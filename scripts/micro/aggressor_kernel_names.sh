cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
for k in qkv d matmul; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/agg_$k -o t -- python $R/scripts/micro/_aggressor_kernels.py $k > /dev/null 2>&1
  f=$(find /tmp/agg_$k -name "*kernel_trace.csv" | head -1)
  echo "== $k"
  grep -o "Cijk[A-Za-z0-9_]*" $f | sort -u
done
