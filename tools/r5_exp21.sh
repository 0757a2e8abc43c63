#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp21.txt -- \
  "b4 default(128)|PG_X=1|$B4" "b4 bigmin 96|PG_BF16_BIG_MIN=96|$B4" "b4 bigmin 64|PG_BF16_BIG_MIN=64|$B4" "b4 eager 4M|PG_EAGER_ADAM_MIN=4194304|$B4" "b4 eager 16M|PG_EAGER_ADAM_MIN=16777216|$B4" "b4 no eager|PG_NO_EAGER_ADAM=1|$B4" \
  "b4 default(128)|PG_X=1|$B4" "b4 bigmin 96|PG_BF16_BIG_MIN=96|$B4" "b4 bigmin 64|PG_BF16_BIG_MIN=64|$B4" "b4 eager 4M|PG_EAGER_ADAM_MIN=4194304|$B4" "b4 eager 16M|PG_EAGER_ADAM_MIN=16777216|$B4" \
  "b4 wg_thr 1e9|PG_WG_THR=1e9|$B4" "b4 wg_thr 16e9|PG_WG_THR=16e9|$B4" "b4 enc lvl 0|PG_ENC_PAR_LEVEL=0|$B4" "b4 enc lvl 2|PG_ENC_PAR_LEVEL=2|$B4" "b4 default(128)|PG_X=1|$B4"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -q -m gpu -k "weight_gradient or wgrad" > gpurun_out/r5/wg_tests2.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r5/wg_tests2.log | cut -c1-200
