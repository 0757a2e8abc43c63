#!/bin/bash
# One GPU-box pass that regenerates the round-5 files kept under profiles/ (run through gpurun, then copy from
# gpurun_out/refresh5/):   gpurun --timeout 2400 -- bash tools/refresh_profiles_r4.sh
O=gpurun_out/refresh5
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-north-star --no-config-legs"
python bench.py --steps 20 --warmup 5 2>$O/cfg1.err | tail -1 > $O/round5_bench_cfg1_b4_f32_1gpu.json          # the driver's command: headline + parity + north_star + cpu_baseline
$B --precision bf16_data --launch-table $O/round5_launch_table_cfg1_b4_bf16_data.txt 2>/dev/null | tail -1 > $O/round5_bench_cfg1_b4_bf16_data_1gpu.json
$B --size 224 --pose_dim 32 --batch 8 2>/dev/null | tail -1 > $O/round5_bench_cfg2_224_p32_b8_f32_1gpu.json
$B --size 224 --pose_dim 32 --batch 8 --precision bf16_data 2>/dev/null | tail -1 > $O/round5_bench_cfg2_224_p32_b8_bf16_data_1gpu.json
$B --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 2>/dev/null | tail -1 > $O/round5_bench_cfg3_nnloss_vgg_b4_f32_1gpu.json
$B --content_loss_layer block1_conv2 --nn_loss_area_size 5 --l1_penalty_weight 0.01 --batch 32 --steps 8 --warmup 3 --precision bf16_data 2>/dev/null | tail -1 > $O/round5_bench_cfg3_nnloss_vgg_b32_bf16_data_1gpu.json
$B --size 512 --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/round5_bench_cfg4_512_b8_f32_1gpu.json
$B --size 512 --batch 8 --steps 10 --warmup 3 --precision bf16_data 2>/dev/null | tail -1 > $O/round5_bench_cfg4_512_b8_bf16_data_1gpu.json
$B --batch 32 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/round5_bench_b32_f32_1gpu.json
$B --batch 32 --steps 10 --warmup 3 --precision bf16_data --launch-table $O/round5_launch_table_b32_bf16_data.txt 2>/dev/null | tail -1 > $O/round5_bench_b32_bf16_data_1gpu.json
# north-star sub-metric: this build, the round-4 configuration of the same build on the same box, and each round-5 change switched off alone
bash tools/ns_ab.sh "" "PG_BIG_MERGE=0 PG_NO_OUT_FWD_FUSED=1 PG_ENC_PAR=0" "PG_BIG_MERGE=0" "PG_NO_OUT_FWD_FUSED=1" "PG_ENC_PAR=0" "PG_PAIR_PERSIST=1" > $O/round5_northstar_gen_fwd_bwd_b32.txt 2>&1
python tools/host_overhead.py f32 2>&1 | tail -3 > $O/round5_host_overhead.txt
python tools/host_overhead.py bf16_data 2>&1 | tail -3 >> $O/round5_host_overhead.txt
python tools/layer_bench.py 32 > $O/round5_layer_bench_b32_bf16.txt 2>/dev/null
echo "---- all-zero operands (same instruction streams: the power / clock limit)" >> $O/round5_layer_bench_b32_bf16.txt
PG_LB_ZERO=1 python tools/layer_bench.py 32 dec3 dec4 dec5 enc3 >> $O/round5_layer_bench_b32_bf16.txt 2>/dev/null
echo "---- fp32 path, batch 4, random / all-zero operands" >> $O/round5_layer_bench_b32_bf16.txt
PG_LB_F32=1 python tools/layer_bench.py 4 dec3 dec4 dec5 enc2 >> $O/round5_layer_bench_b32_bf16.txt 2>/dev/null
PG_LB_F32=1 PG_LB_ZERO=1 python tools/layer_bench.py 4 dec3 dec4 dec5 enc2 >> $O/round5_layer_bench_b32_bf16.txt 2>/dev/null
python tools/pyramid_bench.py > $O/round5_mask_pyramid.txt 2>/dev/null
python tools/warp_bench.py 32 2>/dev/null | grep -v amdgpu > $O/round5_warp_bench.txt
python tools/optim_bench.py 2>/dev/null | grep -v amdgpu > $O/round5_optim_bench.txt
if [ -f pose-transfer_amd/lib/libposegan_hip_timing.so ]; then
  ( export PG_TIMING_EXPERIMENTS=1
    echo "---- timing build (results are wrong, times are not): PG_DEBUG_WARP_BWD 1 = no gather phase, 2 = no candidate search; PG_DEBUG_WARP_FWD 1 = no sampling pass, 2 = no pre-pass work"
    for v in "PG_DEBUG_WARP_BWD=1" "PG_DEBUG_WARP_BWD=2" "PG_DEBUG_WARP_BWD=3" "PG_DEBUG_WARP_FWD=1" "PG_DEBUG_WARP_FWD=2" "PG_DEBUG_WARP_FWD=3"; do
      echo "== [$v]"; env $v python tools/warp_bench.py 32 2>/dev/null | grep "level 0"
    done ) >> $O/round5_warp_bench.txt
fi
# timing experiments of the 256-row kernel (timing library: wrong results, right times)
if [ -f pose-transfer_amd/lib/libposegan_hip_timing.so ]; then
  ( export PG_TIMING_EXPERIMENTS=1 PG_BIG_PAIR=0
    for cfg in "" "PG_DEBUG_HALF_A_FETCH=1" "PG_DEBUG_NO_FETCH=1" "PG_DEBUG_NO_KDMA=1" "PG_DEBUG_NO_KBARRIER=1" "PG_DEBUG_NO_KDMA=1 PG_DEBUG_NO_FETCH=1 PG_DEBUG_NO_KBARRIER=1" "PG_DEBUG_A_EVERY_2ND=1"; do
      echo "== [$cfg] random operands / all-zero operands"
      env $cfg python tools/layer_bench.py 32 dec4 2>/dev/null | grep N=32
      env $cfg PG_LB_ZERO=1 python tools/layer_bench.py 32 dec4 2>/dev/null | grep N=32
    done ) > $O/round5_kloop_experiments_dec4.txt
fi
# rocprofv3 kernel statistics
prof() {  # tag, command...
  tag=$1; shift
  mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- "$@" > $O/prof_$tag/stdout.log 2>&1 || true
  python tools/rocpd_summary.py $(ls $O/prof_$tag/*results.db | head -1) $O/round5_kernel_stats_$tag.csv > /dev/null 2>&1 || true
  if [ -n "$TL" ]; then python tools/timeline_r4.py $(ls $O/prof_$tag/*results.db | head -1) $O/round5_timeline_$tag.txt $TL > /dev/null 2>&1 || true; fi
  rm -rf $O/prof_$tag
}
TL=7 prof default_f32 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile
TL=10 prof northstar_bf16 env PG_ONLY_BF16=1 PG_NS_ITERS=7 python tools/gen_fwd_bwd_bench.py 32
TL=10 prof northstar_bf16_single_stream env PG_ONLY_BF16=1 PG_NO_SIDE_STREAM=1 PG_NS_ITERS=7 python tools/gen_fwd_bwd_bench.py 32
TL=7 prof cfg1_b4_bf16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --precision bf16_data
TL= prof cfg2_224_p32_bf16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --size 224 --pose_dim 32 --batch 8 --precision bf16_data
# PMC passes
bash tools/pmc_bench.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_bench.json $O/round5_pmc.json 2>/dev/null
bash tools/pmc_northstar.sh > $O/pmc_ns.log 2>&1; cp gpurun_out/pmc_northstar.json $O/round5_pmc_northstar.json 2>/dev/null
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_ns
for f in $O/*.json; do echo "$f: $(cut -c1-140 $f)"; done
cat $O/round5_northstar_gen_fwd_bwd_b32.txt
