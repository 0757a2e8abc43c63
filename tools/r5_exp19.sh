#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp19.txt -- \
  "b4 default|PG_X=1|$B4" "b4 wgtr4 128|PG_WGTR4_TARGET=128|$B4" "b4 wgtr4 384|PG_WGTR4_TARGET=384|$B4" "b4 wgtr4 512|PG_WGTR4_TARGET=512|$B4" "b4 wgtr4 1024|PG_WGTR4_TARGET=1024|$B4" \
  "b4 wgtr 64|PG_WGTR_TARGET=64|$B4" "b4 wgtr 256|PG_WGTR_TARGET=256|$B4" "b4 wgtr 512|PG_WGTR_TARGET=512|$B4" \
  "b4 default|PG_X=1|$B4" "b4 bigmin 128|PG_BF16_BIG_MIN=128|$B4" "b4 bigmin 96|PG_BF16_BIG_MIN=96|$B4" "b4 bigmin 256|PG_BF16_BIG_MIN=256|$B4" \
  "b4 splitk fixed 6|PG_SPLITK_FIXED_US=6|$B4" "b4 splitk fixed 20|PG_SPLITK_FIXED_US=20|$B4" "b4 eager 256k|PG_EAGER_ADAM_MIN=262144|$B4" "b4 eager 4M|PG_EAGER_ADAM_MIN=4194304|$B4" "b4 default|PG_X=1|$B4"
