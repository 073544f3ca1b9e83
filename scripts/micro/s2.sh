#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B=scripts/micro/layer_bench
run() { echo "=== $*"; timeout 300 $B --steps 40 "$@" 2>&1 | grep -v "mark\|model\|tau layer"; }
{
run
run --tune gu:16:0:0:0
run --tune down:0:0:2:0
run --tune down:0:0:3:0
run --tune down:16:0:8:0
run --tune qkv:16:0:2:0
run --tune wo:0:0:2:0
run --tune wo:16:0:8:0
run --no_pair
} > gpurun_out/s2.log 2>&1
