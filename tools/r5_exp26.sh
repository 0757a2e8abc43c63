#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
F32B="python bench.py --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 8 --warmup 3 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp26.txt -- \
  "b4 lvl2(default)|PG_X=1|$B4" "b4 lvl3|PG_ENC_PAR_LEVEL=3|$B4" "b4 lvl4|PG_ENC_PAR_LEVEL=4|$B4" "b4 off|PG_ENC_PAR=0|$B4" "b4 lvl2(default)|PG_X=1|$B4" "b4 lvl3|PG_ENC_PAR_LEVEL=3|$B4" \
  "cfg2 lvl2(default)|PG_X=1|$C2" "cfg2 lvl3|PG_ENC_PAR_LEVEL=3|$C2" "cfg2 lvl4|PG_ENC_PAR_LEVEL=4|$C2" \
  "ns lvl1(default)|PG_X=1|$NS" "ns lvl3|PG_ENC_PAR_LEVEL=3|$NS" "ns lvl4|PG_ENC_PAR_LEVEL=4|$NS" "ns off|PG_ENC_PAR=0|$NS" "ns lvl1(default)|PG_X=1|$NS" "ns lvl3|PG_ENC_PAR_LEVEL=3|$NS" "ns lvl4|PG_ENC_PAR_LEVEL=4|$NS" \
  "b32 lvl1(default)|PG_X=1|$B32" "b32 lvl3|PG_ENC_PAR_LEVEL=3|$B32" "b32 lvl4|PG_ENC_PAR_LEVEL=4|$B32" \
  "f32b32 lvl1(default)|PG_X=1|$F32B" "f32b32 lvl4|PG_ENC_PAR_LEVEL=4|$F32B"
