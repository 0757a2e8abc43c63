#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 600 -x -q -k "warp" > gpurun_out/r2/tests_warp.log 2>&1; echo "warp tests rc=$?"; tail -5 gpurun_out/r2/tests_warp.log
for V in 0; do
  if [ $V = 1 ]; then export PG_WARP_FWD_V1=1; else unset PG_WARP_FWD_V1; fi
  PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
  python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([(h['kernel'],h['ms'],h['frac_of_hbm_peak']) for h in d['hbm_kernels'] if 'warp' in h['kernel']])"
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([(h['kernel'],h['ms'],h['frac_of_hbm_peak']) for h in d['hbm_kernels'] if 'warp' in h['kernel']])"
done
