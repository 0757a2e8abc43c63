#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp11.txt -- \
  "ns 128|PG_X=1|$NS" "ns 192|PG_WGTR_TARGET=192|$NS" "ns 160|PG_WGTR_TARGET=160|$NS" "ns 128|PG_X=1|$NS" "ns 192|PG_WGTR_TARGET=192|$NS" "ns 160|PG_WGTR_TARGET=160|$NS" "ns 224|PG_WGTR_TARGET=224|$NS" \
  "b32 128|PG_X=1|$B32" "b32 192|PG_WGTR_TARGET=192|$B32" "b32 128|PG_X=1|$B32" "b32 192|PG_WGTR_TARGET=192|$B32" \
  "b4 128|PG_X=1|$B4" "b4 192|PG_WGTR_TARGET=192|$B4" "b4 128|PG_X=1|$B4" "b4 192|PG_WGTR_TARGET=192|$B4" \
  "cfg2 128|PG_X=1|$C2" "cfg2 192|PG_WGTR_TARGET=192|$C2"
