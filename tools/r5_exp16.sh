#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
tools/r5_ab.sh gpurun_out/r5/exp16.txt -- \
  "b4 128/32|PG_X=1|$B4" "b4 256/32|PG_WGTR_SMALL_WGS=256|$B4" "b4 512/32|PG_WGTR_SMALL_WGS=512|$B4" "b4 256/64|PG_WGTR_SMALL_WGS=256 PG_WGTR_SMALL_KT=64|$B4" "b4 512/128|PG_WGTR_SMALL_WGS=512 PG_WGTR_SMALL_KT=128|$B4" \
  "b4 128/32|PG_X=1|$B4" "b4 256/32|PG_WGTR_SMALL_WGS=256|$B4" "b4 512/32|PG_WGTR_SMALL_WGS=512|$B4" \
  "cfg2 128/32|PG_X=1|$C2" "cfg2 256/32|PG_WGTR_SMALL_WGS=256|$C2" "cfg2 512/64|PG_WGTR_SMALL_WGS=512 PG_WGTR_SMALL_KT=64|$C2" \
  "ns 128/32|PG_X=1|$NS" "ns 256/32|PG_WGTR_SMALL_WGS=256|$NS" "ns 512/64|PG_WGTR_SMALL_WGS=512 PG_WGTR_SMALL_KT=64|$NS"
