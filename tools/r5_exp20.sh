#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B8="python bench.py --precision bf16_data --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B16="python bench.py --precision bf16_data --batch 16 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 50 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp20.txt -- \
  "b4 256|PG_X=1|$B4" "b4 128|PG_WGTR4_TARGET=128|$B4" "b4 96|PG_WGTR4_TARGET=96|$B4" "b4 64|PG_WGTR4_TARGET=64|$B4" "b4 32|PG_WGTR4_TARGET=32|$B4" "b4 256|PG_X=1|$B4" "b4 128|PG_WGTR4_TARGET=128|$B4" "b4 64|PG_WGTR4_TARGET=64|$B4" \
  "cfg2 256|PG_X=1|$C2" "cfg2 128|PG_WGTR4_TARGET=128|$C2" "cfg2 64|PG_WGTR4_TARGET=64|$C2" "cfg2 256|PG_X=1|$C2" "cfg2 128|PG_WGTR4_TARGET=128|$C2" \
  "b8 256|PG_X=1|$B8" "b8 128|PG_WGTR4_TARGET=128|$B8" "b8 64|PG_WGTR4_TARGET=64|$B8" \
  "b16 256|PG_X=1|$B16" "b16 128|PG_WGTR4_TARGET=128|$B16" "b16 192|PG_WGTR4_TARGET=192|$B16"
