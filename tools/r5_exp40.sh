#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
WB="python tools/warp_bench.py 32 2>&1 | grep -E 'level|sum' | cut -c60-110 | tr '\n' ';'"
tools/r5_ab.sh gpurun_out/r5/exp40.txt -- "tiles 256|PG_X=1|$WB" "tiles 128|PG_WARP_BWD_TILES=128|$WB" "tiles 512|PG_WARP_BWD_TILES=512|$WB" "tiles 1024|PG_WARP_BWD_TILES=1024|$WB" "tiles 4096|PG_WARP_BWD_TILES=4096|$WB"
