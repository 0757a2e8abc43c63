#!/bin/bash
# One GPU-box pass that regenerates the round-2 files kept under profiles/ (run through gpurun, then copy from
# gpurun_out/refresh2/):   gpurun --timeout 2400 -- bash tools/refresh_profiles_r2.sh
O=gpurun_out/refresh2
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline"
python bench.py 2>$O/cfg1.err | tail -1 > $O/round2_bench_cfg1_b4_f32_1gpu.json                       # headline: configs[1] incl. cpu_baseline
$B --precision bf16_data --launch-table $O/round2_launch_table_cfg1_b4_bf16_data.txt 2>/dev/null | tail -1 > $O/round2_bench_cfg1_b4_bf16_data_1gpu.json
$B --size 224 --pose_dim 32 --batch 8 --launch-table $O/round2_launch_table_cfg2_f32.txt 2>/dev/null | tail -1 > $O/round2_bench_cfg2_224_p32_b8_f32_1gpu.json
$B --size 224 --pose_dim 32 --batch 8 --precision bf16_data --launch-table $O/round2_launch_table_cfg2_bf16_data.txt 2>/dev/null | tail -1 > $O/round2_bench_cfg2_224_p32_b8_bf16_data_1gpu.json
$B --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 2>/dev/null | tail -1 > $O/round2_bench_cfg3_nnloss_vgg_b4_f32_1gpu.json
$B --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 --batch 32 --steps 5 --warmup 2 --precision bf16_data 2>/dev/null | tail -1 > $O/round2_bench_cfg3_nnloss_vgg_b32_bf16_data_1gpu.json
$B --size 512 --batch 8 --steps 8 2>/dev/null | tail -1 > $O/round2_bench_cfg4_512_b8_f32_1gpu.json
$B --size 512 --batch 8 --steps 8 --precision bf16_data 2>/dev/null | tail -1 > $O/round2_bench_cfg4_512_b8_bf16_data_1gpu.json
$B --batch 32 --steps 8 2>/dev/null | tail -1 > $O/round2_bench_b32_f32_1gpu.json
$B --batch 32 --steps 8 --precision bf16_data --launch-table $O/round2_launch_table_b32_bf16_data.txt 2>/dev/null | tail -1 > $O/round2_bench_b32_bf16_data_1gpu.json
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd" > $O/round2_northstar_gen_fwd_bwd_b32.txt
python tools/host_overhead.py f32 2>&1 | tail -2 > $O/round2_host_overhead.txt
python tools/host_overhead.py bf16_data 2>&1 | tail -2 >> $O/round2_host_overhead.txt
# rocprofv3 kernel statistics: default workload (fp32), north-star workload (bf16), configs[3] (nn-loss + VGG)
prof() {  # tag, command...
  tag=$1; shift
  mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- "$@" > $O/prof_$tag/stdout.log 2>&1 || true
  python tools/rocpd_summary.py $O/prof_$tag/p_results.db $O/round2_kernel_stats_$tag.csv > /dev/null 2>&1 || true
  rm -rf $O/prof_$tag
}
prof default_f32 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile
prof northstar_bf16 env PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32
prof cfg3_nnloss_vgg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01
prof cfg2_224_p32_bf16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --size 224 --pose_dim 32 --batch 8 --precision bf16_data
# PMC passes
bash tools/pmc_bench.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_bench.json $O/round2_pmc.json 2>/dev/null
bash tools/pmc_northstar.sh > $O/pmc_ns.log 2>&1; cp gpurun_out/pmc_northstar.json $O/round2_pmc_northstar.json 2>/dev/null
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_ns
for f in $O/*.json; do echo "$f: $(cut -c1-120 $f)"; done
ls -la $O
