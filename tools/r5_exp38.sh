#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "norm or materialis or determin or bitwise" 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-config-legs --steps 30 > gpurun_out/r5/exp38.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/exp38.json').read().strip().splitlines()[-1])
print(d['value'], d['north_star']['ms'], d['bf16_data_b32_img_s']['value'])
for k in d['north_star']['hbm_kernels']: print(k['kernel'], k['ms'], k['TB_per_s'], k['frac_of_hbm_peak'])
PY
for i in 1 2; do PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 2>/dev/null | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'; done
