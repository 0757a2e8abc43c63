#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp28.txt -- \
  "b4 default|PG_X=1|$B4" "b4 wgtr4 96|PG_WGTR4_TARGET=96|$B4" "b4 wgtr4 160|PG_WGTR4_TARGET=160|$B4" "b4 wgtr4 192|PG_WGTR4_TARGET=192|$B4" "b4 wgtr 64|PG_WGTR_TARGET=64|$B4" "b4 wgtr 256|PG_WGTR_TARGET=256|$B4" \
  "b4 default|PG_X=1|$B4" "b4 big 64|PG_BF16_BIG_MIN=64|$B4" "b4 big 128|PG_BF16_BIG_MIN=128|$B4" "b4 big 48|PG_BF16_BIG_MIN=48|$B4" "b4 wgthr 1e9|PG_WG_THR=1e9|$B4" "b4 wgthr 2e9|PG_WG_THR=2e9|$B4" "b4 smallkt 64|PG_WGTR_SMALL_KT=64|$B4" \
  "b4 default|PG_X=1|$B4" "b4 splitk fixed 6|PG_SPLITK_FIXED_US=6|$B4" "b4 splitk fixed 24|PG_SPLITK_FIXED_US=24|$B4" "b4 splitk launch 12|PG_SPLITK_LAUNCH_US=12|$B4" \
  "f4 default|PG_X=1|$F4" "f4 wg 512|PG_WG_TARGET=512|$F4" "f4 wg 768|PG_WG_TARGET=768|$F4" "f4 splitk fixed 20|PG_SPLITK_FIXED_US=20|$F4" "f4 default|PG_X=1|$F4"
