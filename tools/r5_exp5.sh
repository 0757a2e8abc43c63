#!/bin/bash
# round 5: persistent tap-pair kernel (next tile's prologue under the epilogue) — tests, A/B on the north-star pass
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kernels.py -q -m gpu -k "x_phase or conv_bf16_big_kernel or north_star_shapes" > gpurun_out/r5/pp_tests.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r5/pp_tests.log | cut -c1-300
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp5.txt -- \
  "ns one-tile|PG_PAIR_PERSIST=0|$NS" "ns persistent|PG_X=1|$NS" "ns one-tile|PG_PAIR_PERSIST=0|$NS" "ns persistent|PG_X=1|$NS" \
  "ns persistent 512|PG_PAIR_PERSIST_WGS=512|$NS" "b4 one-tile|PG_PAIR_PERSIST=0|$B4" "b4 persistent|PG_X=1|$B4"
