#!/bin/bash
# bf16 tolerance study: 3 seeds x 2 losses at 256^2, three repeats in the default (atomics) mode,
# one in PG_DETERMINISTIC mode -> gpurun_out/r5/tol_*.log (TOLSTUDY5 lines)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for i in 1 2 3; do
PG_TOL_STUDY=1 timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "bf16_data_step_256 or scalar_gradients" > gpurun_out/r5/tol_$i.log 2>&1; echo "rc $?"
grep -E "TOLSTUDY5|passed|failed" gpurun_out/r5/tol_$i.log | cut -c1-300
done
PG_DETERMINISTIC=1 PG_TOL_STUDY=1 timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "bf16_data_step_256" > gpurun_out/r5/tol_det.log 2>&1; echo "rc $?"
grep -E "TOLSTUDY5|passed|failed" gpurun_out/r5/tol_det.log | cut -c1-300
