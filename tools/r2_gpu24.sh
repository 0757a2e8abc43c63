#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -q -m gpu --timeout 600 -x -q -k "norm or two_training or iteration" 2>&1 | tail -3
for G in 0 32 64 96 192; do
  echo "group $G MB"
  PG_NORM_BWD_GROUP_MB=$G PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd" | cut -c40-125
  PG_NORM_BWD_GROUP_MB=$G python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], [(h['kernel'],h['calls'],h['ms']) for h in d['hbm_kernels'] if 'norm_bwd' in h['kernel']])"
  PG_NORM_BWD_GROUP_MB=$G python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   f32 b32', d['value'], d['ms_per_step'])"
done
