#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for v in 0 1; do
PG_PAIR_PERSIST=$v python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --steps 10 --warmup 3 --launch-table gpurun_out/r5/launch_b32_persist$v.txt > gpurun_out/r5/bench_b32_persist$v.json 2>/dev/null
done
NS="PG_ONLY_BF16=1 PG_NS_ITERS=20 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5/exp6.txt -- \
  "ns one-tile single-stream|PG_PAIR_PERSIST=0 PG_NO_SIDE_STREAM=1|$NS" "ns persistent single-stream|PG_NO_SIDE_STREAM=1|$NS"
