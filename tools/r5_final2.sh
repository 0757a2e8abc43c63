#!/bin/bash
# end of round 5 after the warp-forward change: GPU suite once, smoke, the driver's bench line, and the profile files the change touches
cd $GRAFT_REPO_ROOT; O=gpurun_out/refresh5b; mkdir -p gpurun_out/r5 $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 1800 > gpurun_out/r5/final2_tests.log 2>&1; echo "ALL gpu tests rc=$?"; tail -4 gpurun_out/r5/final2_tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-300
( time python bench.py > $O/round5_bench_cfg1_b4_f32_1gpu.json 2> gpurun_out/r5/bench_final2.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open('gpurun_out/refresh5b/round5_bench_cfg1_b4_f32_1gpu.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['north_star']['ms'], d['bf16_data_b32_img_s']['value'], d['bf16_data_b4_img_s']['value'], d['cfg2_224_p32_b8_bf16']['value'], d['cfg3_nnloss_vgg_b4']['value'], d['roofline']['frac'], d['roofline']['traffic'])
for k in d['north_star']['hbm_kernels']: print(k['kernel'], k['ms'], k['frac_of_hbm_peak'])
PY
python tools/warp_bench.py 32 2>/dev/null | grep -v amdgpu > $O/round5_warp_bench.txt; cat $O/round5_warp_bench.txt
prof() {  # tag, command...
  tag=$1; shift
  mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- "$@" > $O/prof_$tag/stdout.log 2>&1 || true
  python tools/rocpd_summary.py $(ls $O/prof_$tag/*results.db | head -1) $O/round5_kernel_stats_$tag.csv > /dev/null 2>&1 || true
  if [ -n "$TL" ]; then python tools/timeline_r4.py $(ls $O/prof_$tag/*results.db | head -1) $O/round5_timeline_$tag.txt $TL > /dev/null 2>&1 || true; fi
  rm -rf $O/prof_$tag
}
TL=10 prof northstar_bf16 env PG_ONLY_BF16=1 PG_NS_ITERS=7 python tools/gen_fwd_bwd_bench.py 32
TL=10 prof northstar_bf16_single_stream env PG_ONLY_BF16=1 PG_NO_SIDE_STREAM=1 PG_NS_ITERS=7 python tools/gen_fwd_bwd_bench.py 32
TL=7 prof cfg1_b4_bf16 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --precision bf16_data
ls -la $O
