#!/bin/bash
# One GPU-box pass that regenerates the round-3 files kept under profiles/ (run through gpurun, then copy from
# gpurun_out/refresh3/):   gpurun --timeout 2400 -- bash tools/refresh_profiles_r3.sh
O=gpurun_out/refresh3
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline"
python bench.py 2>$O/cfg1.err | tail -1 > $O/round3_bench_cfg1_b4_f32_1gpu.json                       # headline: configs[1] incl. cpu_baseline
$B --precision bf16_data --launch-table $O/round3_launch_table_cfg1_b4_bf16_data.txt 2>/dev/null | tail -1 > $O/round3_bench_cfg1_b4_bf16_data_1gpu.json
$B --precision bf16_data --tape 2>/dev/null | tail -1 > $O/round3_bench_cfg1_b4_bf16_data_tape_1gpu.json
$B --size 224 --pose_dim 32 --batch 8 2>/dev/null | tail -1 > $O/round3_bench_cfg2_224_p32_b8_f32_1gpu.json
$B --size 224 --pose_dim 32 --batch 8 --precision bf16_data --launch-table $O/round3_launch_table_cfg2_bf16_data.txt 2>/dev/null | tail -1 > $O/round3_bench_cfg2_224_p32_b8_bf16_data_1gpu.json
$B --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 2>/dev/null | tail -1 > $O/round3_bench_cfg3_nnloss_vgg_b4_f32_1gpu.json
$B --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 --batch 32 --steps 8 --warmup 3 --precision bf16_data 2>/dev/null | tail -1 > $O/round3_bench_cfg3_nnloss_vgg_b32_bf16_data_1gpu.json
$B --size 512 --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/round3_bench_cfg4_512_b8_f32_1gpu.json
$B --size 512 --batch 8 --steps 10 --warmup 3 --precision bf16_data 2>/dev/null | tail -1 > $O/round3_bench_cfg4_512_b8_bf16_data_1gpu.json
$B --batch 32 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/round3_bench_b32_f32_1gpu.json
$B --batch 32 --steps 10 --warmup 3 --precision bf16_data --launch-table $O/round3_launch_table_b32_bf16_data.txt 2>/dev/null | tail -1 > $O/round3_bench_b32_bf16_data_1gpu.json
PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd" > $O/round3_northstar_gen_fwd_bwd_b32.txt
PG_ONLY_BF16=1 PG_NO_BF16_STORE=1 PG_WGTR4=0 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd" | sed 's/$/   [round-2 configuration: PG_NO_BF16_STORE=1 PG_WGTR4=0, same box]/' >> $O/round3_northstar_gen_fwd_bwd_b32.txt
python tools/host_overhead.py f32 2>&1 | tail -3 > $O/round3_host_overhead.txt
python tools/host_overhead.py bf16_data 2>&1 | tail -3 >> $O/round3_host_overhead.txt
python tools/wgrad_bf16_bench.py > $O/round3_wgrad_bf16_layers.txt 2>/dev/null
echo "---- one-tap kernel only (PG_WGTR4=0)" >> $O/round3_wgrad_bf16_layers.txt
PG_WGTR4=0 python tools/wgrad_bf16_bench.py >> $O/round3_wgrad_bf16_layers.txt 2>/dev/null
# rocprofv3 kernel statistics
prof() {  # tag, command...
  tag=$1; shift
  mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- "$@" > $O/prof_$tag/stdout.log 2>&1 || true
  python tools/rocpd_summary.py $(ls $O/prof_$tag/*results.db | head -1) $O/round3_kernel_stats_$tag.csv > /dev/null 2>&1 || true
  rm -rf $O/prof_$tag
}
prof default_f32 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile
prof northstar_bf16 env PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32
prof northstar_bf16_single_stream env PG_ONLY_BF16=1 PG_NO_SIDE_STREAM=1 python tools/gen_fwd_bwd_bench.py 32
prof cfg2_224_p32_bf16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --size 224 --pose_dim 32 --batch 8 --precision bf16_data
prof cfg1_b4_bf16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --precision bf16_data
# PMC passes
bash tools/pmc_bench.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_bench.json $O/round3_pmc.json 2>/dev/null
bash tools/pmc_northstar.sh > $O/pmc_ns.log 2>&1; cp gpurun_out/pmc_northstar.json $O/round3_pmc_northstar.json 2>/dev/null
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_ns
for f in $O/*.json; do echo "$f: $(cut -c1-140 $f)"; done
cat $O/round3_northstar_gen_fwd_bwd_b32.txt
