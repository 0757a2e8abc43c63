#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "warp_forward_v5" 2>&1 | tail -5
WB="python tools/warp_bench.py 32 2>&1 | grep -E 'level|sum' | cut -c1-58 | tr '\n' ';'"
tools/r5_ab.sh gpurun_out/r5/exp35.txt -- "v3|PG_WARP_FWD_V3=1|$WB" "v5 8192|PG_X=1|$WB" "v5 16384|PG_WARP_FWD5_WGS=16384|$WB" "v5 6144|PG_WARP_FWD5_WGS=6144|$WB"
