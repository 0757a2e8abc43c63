#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py -q -m gpu -k "deterministic or stream or eager or reducer or dp" > gpurun_out/r5/wpar_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r5/wpar_tests.log | cut -c1-250
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp15.txt -- \
  "ns one side|PG_NO_WGRAD_PAR=1|$NS" "ns two sides|PG_X=1|$NS" "ns one side|PG_NO_WGRAD_PAR=1|$NS" "ns two sides|PG_X=1|$NS" "ns one side|PG_NO_WGRAD_PAR=1|$NS" "ns two sides|PG_X=1|$NS" \
  "b32 one|PG_NO_WGRAD_PAR=1|$B32" "b32 two|PG_X=1|$B32" "b4 one|PG_NO_WGRAD_PAR=1|$B4" "b4 two|PG_X=1|$B4" "b4 one|PG_NO_WGRAD_PAR=1|$B4" "b4 two|PG_X=1|$B4" \
  "f4 one|PG_NO_WGRAD_PAR=1|$F4" "f4 two|PG_X=1|$F4" "f4 one|PG_NO_WGRAD_PAR=1|$F4" "f4 two|PG_X=1|$F4"
