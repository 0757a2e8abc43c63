#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp29.txt -- \
  "ns default|PG_X=1|$NS" "ns hwq 8|GPU_MAX_HW_QUEUES=8|$NS" "ns hwq 2|GPU_MAX_HW_QUEUES=2|$NS" "ns hwq 6|GPU_MAX_HW_QUEUES=6|$NS" "ns default|PG_X=1|$NS" "ns hwq 8|GPU_MAX_HW_QUEUES=8|$NS" \
  "b4 default|PG_X=1|$B4" "b4 hwq 8|GPU_MAX_HW_QUEUES=8|$B4" "b4 hwq 2|GPU_MAX_HW_QUEUES=2|$B4" "b32 default|PG_X=1|$B32" "b32 hwq 8|GPU_MAX_HW_QUEUES=8|$B32" "f4 default|PG_X=1|$F4" "f4 hwq 8|GPU_MAX_HW_QUEUES=8|$F4"
