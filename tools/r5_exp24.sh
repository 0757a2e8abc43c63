#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp24.txt -- \
  "f4 default|PG_X=1|$F4" "f4 wg 512|PG_WG_TARGET=512|$F4" "f4 wg 768|PG_WG_TARGET=768|$F4" "f4 wg 1536|PG_WG_TARGET=1536|$F4" "f4 wg 2048|PG_WG_TARGET=2048|$F4" \
  "f4 default|PG_X=1|$F4" "f4 splitk fixed 6|PG_SPLITK_FIXED_US=6|$F4" "f4 splitk fixed 20|PG_SPLITK_FIXED_US=20|$F4" "f4 splitk bw 2|PG_SPLITK_BW_TBS=2|$F4" "f4 splitk bw 5|PG_SPLITK_BW_TBS=5|$F4" "f4 launch 3|PG_SPLITK_LAUNCH_US=3|$F4" "f4 launch 12|PG_SPLITK_LAUNCH_US=12|$F4" \
  "f4 default|PG_X=1|$F4" "f4 no prefetch|PG_NO_GEN_PREFETCH=1|$F4" "f4 no aux|PG_NO_AUX_STREAM=1|$F4" "f4 enc off|PG_ENC_PAR=0|$F4" "f4 enc lvl 3|PG_ENC_PAR_LEVEL=3|$F4" "f4 default|PG_X=1|$F4"
