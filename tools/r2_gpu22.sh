#!/bin/bash
export TMPDIR=/tmp
for T in 1024 512 256 2048; do
  echo "target $T"
  for A in "" "--batch 32 --steps 5 --warmup 2" "--size 224 --pose_dim 32 --batch 8"; do
  PG_WG_TARGET=$T python bench.py $A --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'])"
  done
done
