#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
OLD="PG_WGTR_SMALL_WGS=128 PG_WGTR_NO_SMALL_64=1"
tools/r5_ab.sh gpurun_out/r5/exp18.txt -- \
  "b4 old|$OLD|$B4" "b4 new|PG_X=1|$B4" "b4 kt64|PG_WGTR_SMALL_KT=64|$B4" "b4 kt128|PG_WGTR_SMALL_KT=128|$B4" "b4 old|$OLD|$B4" "b4 new|PG_X=1|$B4" "b4 kt64|PG_WGTR_SMALL_KT=64|$B4" \
  "cfg2 old|$OLD|$C2" "cfg2 new|PG_X=1|$C2" "cfg2 kt64|PG_WGTR_SMALL_KT=64|$C2" "cfg2 old|$OLD|$C2" "cfg2 new|PG_X=1|$C2" \
  "b32 old|$OLD|$B32" "b32 new|PG_X=1|$B32" "b32 kt64|PG_WGTR_SMALL_KT=64|$B32" "b32 old|$OLD|$B32" "b32 new|PG_X=1|$B32"
