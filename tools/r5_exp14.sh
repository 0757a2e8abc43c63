#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
PG_TOL_STUDY=1 timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -q -s -m gpu -k "512_vs_golden or 256_vs_golden" > gpurun_out/r5/g512_tests.log 2>&1; echo "rc $?"
grep -aE "TOLSTUDY5|passed|failed|Error|assert" gpurun_out/r5/g512_tests.log | cut -c1-300 | tail -30
