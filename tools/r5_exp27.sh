#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp27.txt -- \
  "b4 default(off)|PG_X=1|$B4" "b4 no aux|PG_NO_AUX_STREAM=1|$B4" "b4 no prefetch|PG_NO_GEN_PREFETCH=1|$B4" "b4 no side|PG_NO_SIDE_STREAM=1|$B4" "b4 tape|PG_X=1|${B4/--steps 150/--steps 150 --tape}" "b4 default(off)|PG_X=1|$B4" "b4 no aux|PG_NO_AUX_STREAM=1|$B4" "b4 no prefetch|PG_NO_GEN_PREFETCH=1|$B4" \
  "f4 default(off)|PG_X=1|$F4" "f4 no aux|PG_NO_AUX_STREAM=1|$F4" "f4 no prefetch|PG_NO_GEN_PREFETCH=1|$F4" "f4 default(off)|PG_X=1|$F4" \
  "cfg2 default(off)|PG_X=1|$C2" "cfg2 lvl4|PG_ENC_PAR_LEVEL=4|$C2" "cfg2 no aux|PG_NO_AUX_STREAM=1|$C2" "cfg2 default(off)|PG_X=1|$C2"
