#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_networks.py tests/test_gpu_round2.py -q -m gpu -x > gpurun_out/r5/final3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5/final3_tests.log | cut -c1-200
bash tools/refresh_profiles_r5.sh > gpurun_out/r5/refresh3.log 2>&1; tail -14 gpurun_out/r5/refresh3.log | cut -c1-150
