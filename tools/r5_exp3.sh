#!/bin/bash
# round 5: the one-pass output convolution forward — its tests, then A/B on the north-star pass and the batch-32 / batch-4 steps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -m gpu -k "out_conv_fwd" > gpurun_out/r5/ocf_tests.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r5/ocf_tests.log
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -q -x -m gpu > gpurun_out/r5/ocf_tests2.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r5/ocf_tests2.log
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5/exp3.txt -- \
  "ns unfused|PG_NO_OUT_FWD_FUSED=1|$NS" "ns fused|PG_X=1|$NS" "ns unfused|PG_NO_OUT_FWD_FUSED=1|$NS" "ns fused|PG_X=1|$NS" \
  "b32 unfused|PG_NO_OUT_FWD_FUSED=1|$B32" "b32 fused|PG_X=1|$B32" "b4 unfused|PG_NO_OUT_FWD_FUSED=1|$B4" "b4 fused|PG_X=1|$B4"
