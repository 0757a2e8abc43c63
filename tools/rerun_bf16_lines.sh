O=gpurun_out/refresh
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --precision bf16 2>/dev/null | tail -1 > $O/round1_bench_cfg3_bf16_operands_1gpu.json
python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_cfg3_bf16_data_1gpu.json
python bench.py --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_cfg2shape_bf16_data_1gpu.json
python bench.py --size 512 --batch 8 --steps 8 --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_cfg5_bf16_data_1gpu.json
python bench.py --batch 32 --steps 8 --no-cpu-baseline --precision bf16 2>/dev/null | tail -1 > $O/round1_bench_b32_bf16_operands_1gpu.json
python bench.py --batch 32 --steps 8 --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 > $O/round1_bench_b32_bf16_data_1gpu.json
