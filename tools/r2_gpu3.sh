#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/microbench/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe > gpurun_out/r2/tr_probe.txt 2>&1; head -40 gpurun_out/r2/tr_probe.txt
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bf16_big or bf16_data" --timeout 900 > gpurun_out/r2/t_big.log 2>&1; echo "big kernel tests rc=$?"; tail -5 gpurun_out/r2/t_big.log
python -m pytest tests/test_gpu_round2.py -q -m gpu --timeout 1500 > gpurun_out/r2/tests_round2.log 2>&1; echo "round2 rc=$?"; tail -25 gpurun_out/r2/tests_round2.log
