#!/bin/bash
# round 5: x-phase merged transposed convolutions — tests, A/B on the north-star pass; per-kernel times of the batch-32 step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu -k "out_conv_fwd or x_phase" > gpurun_out/r5/mg_tests.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r5/mg_tests.log | cut -c1-300
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5/exp4.txt -- \
  "ns unmerged|PG_BIG_MERGE=0|$NS" "ns merged|PG_X=1|$NS" "ns unmerged|PG_BIG_MERGE=0|$NS" "ns merged|PG_X=1|$NS"
python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --steps 20 --launch-table gpurun_out/r5/launch_b32_merged.txt > gpurun_out/r5/bench_b32_merged.json 2> gpurun_out/r5/bench_b32_merged.err; tail -c 600 gpurun_out/r5/bench_b32_merged.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_b32_merged.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d.get('hbm_kernels', []): print(k['kernel'], k['calls'], k['ms'], k.get('TB_per_s'))
PY
