#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py -q -m gpu -k "deterministic or stream or eager or tape or graph or replay" > gpurun_out/r5/sched_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5/sched_tests.log | cut -c1-250
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C2="python bench.py --precision bf16_data --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 80 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
C3="python bench.py --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
OLD="PG_EAGER_ADAM=1 PG_BF16_BIG_MIN=192 PG_ENC_PAR_LEVEL=1 PG_WGTR4_TARGET=256 PG_WGTR_SMALL_WGS=128 PG_WGTR_NO_SMALL_64=1"
tools/r5_ab.sh gpurun_out/r5/exp23.txt -- \
  "b4 before|$OLD|$B4" "b4 now|PG_X=1|$B4" "b4 before|$OLD|$B4" "b4 now|PG_X=1|$B4" \
  "f4 before|$OLD|$F4" "f4 now|PG_X=1|$F4" "f4 before|$OLD|$F4" "f4 now|PG_X=1|$F4" \
  "cfg2 before|$OLD|$C2" "cfg2 now|PG_X=1|$C2" "cfg3 before|$OLD|$C3" "cfg3 now|PG_X=1|$C3" \
  "b32 before|$OLD|$B32" "b32 now|PG_X=1|$B32" "ns before|$OLD|$NS" "ns now|PG_X=1|$NS" "ns before|$OLD|$NS" "ns now|PG_X=1|$NS"
