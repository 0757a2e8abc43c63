#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NE="PG_NO_EAGER_ADAM=1"
tools/r5_ab.sh gpurun_out/r5/exp22.txt -- \
  "b4 default|PG_X=1|$B4" "b4 noeager|$NE|$B4" "b4 noeager+big96|$NE PG_BF16_BIG_MIN=96|$B4" "b4 noeager+big96+lvl2|$NE PG_BF16_BIG_MIN=96 PG_ENC_PAR_LEVEL=2|$B4" "b4 noeager+lvl2|$NE PG_ENC_PAR_LEVEL=2|$B4" "b4 default|PG_X=1|$B4" "b4 noeager+big96+lvl2|$NE PG_BF16_BIG_MIN=96 PG_ENC_PAR_LEVEL=2|$B4" \
  "f4 default|PG_X=1|$F4" "f4 noeager|$NE|$F4" "f4 noeager+lvl2|$NE PG_ENC_PAR_LEVEL=2|$F4" "f4 default|PG_X=1|$F4" "f4 noeager|$NE|$F4" "f4 lvl2|PG_ENC_PAR_LEVEL=2|$F4" \
  "cfg2 default|PG_X=1|$C2" "cfg2 noeager|$NE|$C2" "cfg2 noeager+big96|$NE PG_BF16_BIG_MIN=96|$C2" "cfg2 default|PG_X=1|$C2" "cfg2 noeager|$NE|$C2" \
  "b32 default|PG_X=1|$B32" "b32 noeager|$NE|$B32" "b32 big96|PG_BF16_BIG_MIN=96|$B32" "b32 default|PG_X=1|$B32" "b32 noeager|$NE|$B32"
