#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 600 -x -q -k "big_kernel" > gpurun_out/r2/tests_big.log 2>&1; echo "big tests rc=$?"; tail -5 gpurun_out/r2/tests_big.log
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
PG_NO_BF16_BIG64=1 PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --launch-table gpurun_out/r2/lt_bf16_b32.txt > gpurun_out/r2/bench_bf16_b32.json 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_bf16_b32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print({k:(v['ms'],v['tflops']) for k,v in d['roofline']['families'].items()})
PY
