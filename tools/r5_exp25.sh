#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C3="python bench.py --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp25.txt -- \
  "f4 default(lvl1)|PG_X=1|$F4" "f4 enc off|PG_ENC_PAR=0|$F4" "f4 lvl 2|PG_ENC_PAR_LEVEL=2|$F4" "f4 lvl 3|PG_ENC_PAR_LEVEL=3|$F4" "f4 lvl 4|PG_ENC_PAR_LEVEL=4|$F4" "f4 lvl 5|PG_ENC_PAR_LEVEL=5|$F4" \
  "f4 default(lvl1)|PG_X=1|$F4" "f4 enc off|PG_ENC_PAR=0|$F4" "f4 lvl 3|PG_ENC_PAR_LEVEL=3|$F4" "f4 lvl 4|PG_ENC_PAR_LEVEL=4|$F4" \
  "f4 lvl3 noprefetch|PG_ENC_PAR_LEVEL=3 PG_NO_GEN_PREFETCH=1|$F4" "f4 encoff noprefetch|PG_ENC_PAR=0 PG_NO_GEN_PREFETCH=1|$F4" "f4 lvl3 eager|PG_ENC_PAR_LEVEL=3 PG_EAGER_ADAM=1|$F4" \
  "cfg3 default|PG_X=1|$C3" "cfg3 lvl 3|PG_ENC_PAR_LEVEL=3|$C3" "cfg3 enc off|PG_ENC_PAR=0|$C3"
