#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --timeout 1800 -x -q > gpurun_out/r2/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -6 gpurun_out/r2/tests_all.log
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
PG_ONLY_BF16=1 PG_NO_XCD_SWIZZLE=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --launch-table gpurun_out/r2/lt_bf16_b32.txt > gpurun_out/r2/bench_bf16_b32.json 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_bf16_b32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print({k:(v['ms'],v['tflops']) for k,v in d['roofline']['families'].items()})
PY
bash tools/pmc_northstar.sh 2>&1 | head -3 | cut -c1-330
