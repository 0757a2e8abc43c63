#!/bin/bash
# One GPU-box pass that regenerates everything kept under profiles/ (run through gpurun, then copy from gpurun_out/:
# the bench lines as they are, the rocprofv3 databases gpurun_out/prof_{refresh,bf,ss}/bench_results.db through
# tools/rocpd_summary.py, gpurun_out/pmc_bench.json as round1_pmc.json):
#   gpurun --timeout 1500 -- bash tools/refresh_profiles.sh
mkdir -p gpurun_out/refresh
O=gpurun_out/refresh
python bench.py 2>$O/cfg2.err | tail -1 > $O/round1_bench_cfg2_1gpu.json
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/round1_bench_cfg3_1gpu.json
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --precision bf16 2>/dev/null | tail -1 > $O/round1_bench_cfg3_bf16_operands_1gpu.json
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_cfg3_bf16_data_1gpu.json
python bench.py --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_cfg2shape_bf16_data_1gpu.json
python bench.py --size 512 --batch 8 --steps 8 --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_cfg5_bf16_data_1gpu.json
python bench.py --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 --no-cpu-baseline 2>/dev/null | tail -1 > $O/round1_bench_cfg4_1gpu.json
python bench.py --size 512 --batch 8 --steps 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/round1_bench_cfg5_1gpu.json
python bench.py --batch 32 --steps 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/round1_bench_b32_1gpu.json
python bench.py --batch 32 --steps 8 --no-cpu-baseline --precision bf16x3 2>/dev/null | tail -1 > $O/round1_bench_b32_bf16x3_operands_1gpu.json
python bench.py --batch 32 --steps 8 --no-cpu-baseline --precision bf16 2>/dev/null | tail -1 > $O/round1_bench_b32_bf16_operands_1gpu.json
python bench.py --batch 32 --steps 8 --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_b32_bf16_data_1gpu.json
bash tools/profile_bench_bf16.sh > $O/profile_bf16.log 2>&1
bash tools/profile_bench.sh refresh > $O/profile.log 2>&1
PG_NO_SIDE_STREAM=1 bash tools/profile_bench.sh ss > $O/profile_single_stream.log 2>&1   # -> round1_kernel_stats_single_stream.csv
bash tools/pmc_bench.sh > $O/pmc.log 2>&1
for f in $O/*.json; do echo "$f: $(cut -c1-150 $f)"; done
