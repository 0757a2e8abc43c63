#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
O=gpurun_out/r2
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 600 -x -q -k "small_cin_patch_wgrad" > $O/tests_scw.log 2>&1; echo "small_cin wgrad tests rc=$?"; tail -3 $O/tests_scw.log
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --launch-table $O/lt_cfg3_f32.txt 2>/dev/null | tail -1 > $O/bench_cfg3_f32.json
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --precision bf16_data --launch-table $O/lt_cfg3_bf16.txt 2>/dev/null | tail -1 > $O/bench_cfg3_bf16.json
python - <<'PY'
import json
for f in ('gpurun_out/r2/bench_cfg3_f32.json','gpurun_out/r2/bench_cfg3_bf16.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step']); print({k:(v['ms'],v['tflops']) for k,v in d['roofline']['families'].items()})
PY
bash tools/profile_northstar.sh
python tools/rocpd_summary.py gpurun_out/prof_ns/ns_results.db 2>/dev/null | head -40
