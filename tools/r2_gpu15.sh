#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -q -m gpu --timeout 600 -x -q -k "nn_loss or vgg or step_nn or nn" > gpurun_out/r2/tests_nn.log 2>&1; echo "nn/vgg tests rc=$?"; tail -5 gpurun_out/r2/tests_nn.log
for args in "--batch 32 --steps 5 --warmup 2 --precision bf16_data" ""; do
python bench.py --no-cpu-baseline --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([(h['kernel'],h['calls'],h['ms'],h['frac_of_hbm_peak']) for h in d['hbm_kernels'] if 'nn' in h['kernel'] or 'vgg' in h['kernel']])"
done
