#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 2>/dev/null | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
B32="python bench.py --precision bf16_data --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp36.txt -- "ns v3|PG_WARP_FWD_V3=1|$NS" "ns v5|PG_X=1|$NS" "ns v3|PG_WARP_FWD_V3=1|$NS" "ns v5|PG_X=1|$NS" "b32 v3|PG_WARP_FWD_V3=1|$B32" "b32 v5|PG_X=1|$B32" "b4 v3|PG_WARP_FWD_V3=1|$B4" "b4 v5|PG_X=1|$B4" "b4 v3|PG_WARP_FWD_V3=1|$B4" "b4 v5|PG_X=1|$B4"
