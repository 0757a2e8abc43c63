#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu --timeout 600 -x -q -k "hip_graph" 2>&1 | tail -25
for A in "--precision bf16_data" ""; do
  for G in "" "--graph"; do
    timeout 600 python bench.py --no-cpu-baseline --no-kernel-profile $A $G 2>&1 | tail -1 | cut -c1-250
  done
done
