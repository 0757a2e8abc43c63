#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "warp or deform or skip or determin or overflow or gather" 2>&1 | tail -3
for i in 1 2; do PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 2>/dev/null | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'; done
python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b4', d['value'], d['ms_per_step'])"
python bench.py --precision bf16_data --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32', d['value'], d['ms_per_step'])"
