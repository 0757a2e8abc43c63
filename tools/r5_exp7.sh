#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
echo "== default"; python tools/layer_bench.py 32 enc1 enc2 enc3 dec5 2>/dev/null | grep -v amdgpu
echo "== PG_NO_BF16_BIG=1 (generic 128-row kernels)"; PG_NO_BF16_BIG=1 python tools/layer_bench.py 32 enc1 enc2 enc3 2>/dev/null | grep -v amdgpu
echo "== PG_BIG_MERGE=0"; PG_BIG_MERGE=0 python tools/layer_bench.py 32 enc1 enc2 dec5 2>/dev/null | grep -v amdgpu
