#!/bin/bash
# session 1: baseline phases + waves sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
B=scripts/micro/layer_bench
T4="qkv:0:4:0:8,wo:0:4:0:8,gu:0:4:0:8,down:0:4:0:8"
T44="qkv:0:4:0:4,wo:0:4:0:4,gu:0:4:0:4,down:0:4:0:4"
T8="qkv:0:8:0:4,wo:0:8:0:4,gu:0:8:0:4,down:0:8:0:4"
T88="qkv:0:8:0:8,wo:0:8:0:8,gu:0:8:0:8,down:0:8:0:8"
{
echo "=== baseline"; timeout 300 $B --steps 60 --phase
echo "=== dense"; timeout 300 $B --steps 40 --dense
echo "=== waves8 u4"; timeout 300 $B --steps 60 --phase --tune $T8
echo "=== waves8 u8"; timeout 300 $B --steps 60 --phase --tune $T88
echo "=== waves4 u8"; timeout 300 $B --steps 60 --phase --tune $T4
echo "=== waves4 u4"; timeout 300 $B --steps 60 --tune $T44
echo "=== gu only waves4 u8"; timeout 300 $B --steps 60 --tune gu:0:4:0:8
echo "=== gu only waves8 u8"; timeout 300 $B --steps 60 --tune gu:0:8:0:8
echo "=== pos 1500"; timeout 300 $B --steps 40 --pos 1500
} > gpurun_out/s1.log 2>&1


tail -5 gpurun_out/s1.log
