#!/bin/bash
export TMPDIR=/tmp
for T in 768 384 256 128 64; do
  echo "target $T"
  PG_WGTR_TARGET=$T PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd" | cut -c40-120
  for A in "--batch 32 --steps 5 --warmup 2" "" "--size 224 --pose_dim 32 --batch 8"; do
  PG_WGTR_TARGET=$T python bench.py --precision bf16_data $A --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"
  done
done
