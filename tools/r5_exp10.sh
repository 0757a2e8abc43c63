#!/bin/bash
# round 5: re-sweep of the weight-gradient split targets and the encoder-stream level on the north-star pass (the kernels next to them changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
tools/r5_ab.sh gpurun_out/r5/exp10.txt -- \
  "default|PG_X=1|$NS" "wgtr 64|PG_WGTR_TARGET=64|$NS" "wgtr 192|PG_WGTR_TARGET=192|$NS" "wgtr 256|PG_WGTR_TARGET=256|$NS" \
  "wgtr4 128|PG_WGTR4_TARGET=128|$NS" "wgtr4 384|PG_WGTR4_TARGET=384|$NS" "wgtr4 512|PG_WGTR4_TARGET=512|$NS" \
  "default|PG_X=1|$NS" "enc lvl 2|PG_ENC_PAR_LEVEL=2|$NS" "enc lvl 3|PG_ENC_PAR_LEVEL=3|$NS" "enc lvl 0|PG_ENC_PAR_LEVEL=0|$NS" \
  "no aux|PG_NO_AUX_STREAM=1|$NS" "big min 128|PG_BF16_BIG_MIN=128|$NS" "big min 256|PG_BF16_BIG_MIN=256|$NS" "default|PG_X=1|$NS"
