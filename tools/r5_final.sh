#!/bin/bash
# end-of-round check: the whole GPU suite twice (flakiness), smoke, the driver's default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for i in 1 2; do
timeout 2400 python -m pytest tests -q -m gpu --timeout 1800 > gpurun_out/r5/final_tests_$i.log 2>&1; echo "ALL gpu tests run $i rc=$?"; tail -4 gpurun_out/r5/final_tests_$i.log | cut -c1-300
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-300
( time python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['north_star']['ms'], d['bf16_data_b32_img_s']['value'], d['bf16_data_b4_img_s']['value'], d['cfg2_224_p32_b8_bf16']['value'], d['cfg3_nnloss_vgg_b4']['value'], d['roofline']['frac'], d['roofline']['traffic'])
PY
