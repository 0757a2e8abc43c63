#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for i in 4 5 6; do
PG_TOL_STUDY=1 timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -m gpu -k "bf16_data_step_256 or scalar_gradients" > gpurun_out/r5/tol_$i.log 2>&1; echo "rc $?"
grep -aE "TOLSTUDY5" gpurun_out/r5/tol_$i.log | sed 's/^[.F]*//' | cut -c1-330
done
timeout 2400 python -m pytest tests -q -m gpu --timeout 1800 --deselect tests/test_gpu_round5.py::test_bf16_data_step_256_vs_reference --deselect tests/test_gpu_round5.py::test_fp32_step_256_scalar_gradients_vs_golden > gpurun_out/r5/tests_all2.log 2>&1; echo "ALL gpu tests rc=$?"; tail -8 gpurun_out/r5/tests_all2.log | cut -c1-300
