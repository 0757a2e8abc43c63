#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for i in 1 2; do
PG_TRAJ_PRINT=1 timeout 1500 python -m pytest tests/test_gpu_round4.py -q -s -m gpu -k trains_like > gpurun_out/r5/traj_det$i.log 2>&1; echo "traj rc $?"
grep -aE "TRAJ|passed|failed|Error" gpurun_out/r5/traj_det$i.log | cut -c1-330
done
