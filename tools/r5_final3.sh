#!/bin/bash
# end of round 5 after the warp-backward change: GPU suite once, smoke, the driver's bench line, warp bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/refresh5c; mkdir -p gpurun_out/r5 $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --timeout 1200 > gpurun_out/r5/final4_tests.log 2>&1; echo "ALL gpu tests rc=$?"; tail -3 gpurun_out/r5/final4_tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-100
( time python bench.py > $O/round5_bench_cfg1_b4_f32_1gpu.json 2> gpurun_out/r5/bench_final3.err ) 2>&1 | tail -4 | head -2
python - <<'PY'
import json
d=json.loads(open('gpurun_out/refresh5c/round5_bench_cfg1_b4_f32_1gpu.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['north_star']['ms'], d['bf16_data_b32_img_s']['value'], d['bf16_data_b4_img_s']['value'], d['cfg2_224_p32_b8_bf16']['value'], d['cfg3_nnloss_vgg_b4']['value'], d['roofline']['frac'], d['roofline']['traffic'])
for k in d['north_star']['hbm_kernels']: print(k['kernel'], k['ms'], k['frac_of_hbm_peak'])
PY
python tools/warp_bench.py 32 2>/dev/null | grep -v amdgpu > $O/round5_warp_bench.txt; cat $O/round5_warp_bench.txt
