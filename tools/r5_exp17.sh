#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 150 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
tools/r5_ab.sh gpurun_out/r5/exp17.txt -- \
  "b4 128|PG_WGTR_SMALL_WGS=128|$B4" "b4 512|PG_X=1|$B4" "b4 512+64|PG_WGTR_SMALL_64=1|$B4" "b4 1024+64|PG_WGTR_SMALL_64=1 PG_WGTR_SMALL_WGS=1024|$B4" \
  "b4 128|PG_WGTR_SMALL_WGS=128|$B4" "b4 512|PG_X=1|$B4" "b4 512+64|PG_WGTR_SMALL_64=1|$B4" "b4 1024+64|PG_WGTR_SMALL_64=1 PG_WGTR_SMALL_WGS=1024|$B4" \
  "b4 128|PG_WGTR_SMALL_WGS=128|$B4" "b4 512|PG_X=1|$B4"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -q -m gpu -k "weight_gradient or wgrad" > gpurun_out/r5/wg_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5/wg_tests.log | cut -c1-200
