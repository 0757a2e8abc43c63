#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py tests/test_gpu_round2.py -q -m gpu --timeout 900 -x -q -k "stem or bf16 or p32_step" 2>&1 | tail -4
for E in "" "PG_NO_STEM_EMIT_BF16=1"; do
  env $E PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
  env $E python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([(h['kernel'],h['calls'],h['ms'],h['frac_of_hbm_peak']) for h in d['hbm_kernels'] if 'stem' in h['kernel'] or 'mater' in h['kernel']])"
done
