#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B8="python bench.py --precision bf16_data --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B16="python bench.py --precision bf16_data --batch 16 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 50 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B12="python bench.py --precision bf16_data --batch 12 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 50 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C4="python bench.py --precision bf16_data --size 512 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 10 --warmup 3 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp31.txt -- \
  "b8 default(off)|PG_X=1|$B8" "b8 lvl1|PG_ENC_PAR_LEVEL=1|$B8" "b8 lvl4|PG_ENC_PAR_LEVEL=4|$B8" "b8 default(off)|PG_X=1|$B8" \
  "b12 default(off)|PG_X=1|$B12" "b12 lvl1|PG_ENC_PAR_LEVEL=1|$B12" "b12 wgtr4 256|PG_WGTR4_TARGET=256|$B12" \
  "b16 default(lvl1)|PG_X=1|$B16" "b16 off|PG_ENC_PAR=0|$B16" "b16 lvl4|PG_ENC_PAR_LEVEL=4|$B16" "b16 default(lvl1)|PG_X=1|$B16" "b16 off|PG_ENC_PAR=0|$B16" \
  "512b8 default(off)|PG_X=1|$C4" "512b8 lvl1|PG_ENC_PAR_LEVEL=1|$C4" "512b8 wgtr4 256|PG_WGTR4_TARGET=256|$C4" "512b8 default(off)|PG_X=1|$C4"
