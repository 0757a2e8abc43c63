#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "warp or transposing or bf16" --timeout 900 > gpurun_out/r2/t_k.log 2>&1; echo "kernel tests rc=$?"; tail -6 gpurun_out/r2/t_k.log
python -m pytest tests/test_gpu_round2.py -q -m gpu -k "full_warp or stacked or 224 or 512" --timeout 1500 > gpurun_out/r2/tests_net.log 2>&1; echo "round2 subset rc=$?"; tail -5 gpurun_out/r2/tests_net.log
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --launch-table gpurun_out/r2/lt_bf16_b32.txt > gpurun_out/r2/bench_bf16_b32.json 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_bf16_b32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print({k:(v['ms'],v['tflops']) for k,v in d['roofline']['families'].items()})
print([(h['kernel'],h['ms'],h['frac_of_hbm_peak']) for h in d['hbm_kernels'][:8]])
PY
bash tools/pmc_northstar.sh 2>&1 | head -4 | cut -c1-330
