#!/bin/bash
# One GPU-box pass that regenerates the round-6 files kept under profiles/ (run through gpurun, then copy from gpurun_out/refresh6/):
#   gpurun --timeout 2700 -- bash tools/refresh_profiles_r6.sh
# What is new against round 5: the rocprofv3 --kernel-trace --stats summary of the DEFAULT fp32 step is taken
# SINGLE-STREAM (PG_NO_SIDE_STREAM=1: no weight-gradient side stream, hence no auxiliary / second-encoder / prefetch stream), so that a
# kernel's average duration is that kernel alone and `roofline.frac` can be recomputed from the CSV; the multi-stream CSV sits next to it.
O=gpurun_out/refresh6
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-extra-legs"
python bench.py --steps 20 --warmup 5 2>$O/cfg1.err | tail -1 > $O/round6_bench_cfg1_b4_f32_1gpu.json          # the driver's command
$B --precision bf16_data --launch-table $O/round6_launch_table_cfg1_b4_bf16_data.txt 2>/dev/null | tail -1 > $O/round6_bench_cfg1_b4_bf16_data_1gpu.json
$B --batch 32 --steps 10 --warmup 3 --precision bf16_data --launch-table $O/round6_launch_table_b32_bf16_data.txt 2>/dev/null | tail -1 > $O/round6_bench_b32_bf16_data_1gpu.json
$B --size 512 --batch 8 --steps 10 --warmup 3 --precision bf16_data 2>/dev/null | tail -1 > $O/round6_bench_cfg4_512_b8_bf16_data_1gpu.json
python tools/host_overhead.py f32 2>&1 | tail -3 > $O/round6_host_overhead.txt
python tools/host_overhead.py bf16_data 2>&1 | tail -3 >> $O/round6_host_overhead.txt
python tools/layer_bench.py 32 2>/dev/null | grep -v amdgpu > $O/round6_layer_bench_b32_bf16.txt
python tools/warp_bench.py 32 2>/dev/null | grep -v amdgpu > $O/round6_warp_bench.txt
# rocprofv3 kernel statistics
prof() {  # tag, command...
  tag=$1; shift
  mkdir -p $O/prof_$tag
  rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- "$@" > $O/prof_$tag/stdout.log 2>&1 || true
  python tools/rocpd_summary.py $(ls $O/prof_$tag/*results.db | head -1) $O/round6_kernel_stats_$tag.csv > /dev/null 2>&1 || true
  if [ -n "$TL" ]; then python tools/timeline_r4.py $(ls $O/prof_$tag/*results.db | head -1) $O/round6_timeline_$tag.txt $TL > /dev/null 2>&1 || true; fi
  rm -rf $O/prof_$tag
}
PB="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --no-config-legs --no-extra-legs --no-kernel-profile"
TL=7 prof default_f32_single_stream env PG_NO_SIDE_STREAM=1 $PB
TL=7 prof default_f32 $PB
TL=10 prof northstar_bf16 env PG_ONLY_BF16=1 PG_NS_ITERS=7 python tools/gen_fwd_bwd_bench.py 32
TL=10 prof northstar_bf16_single_stream env PG_ONLY_BF16=1 PG_NO_SIDE_STREAM=1 PG_NS_ITERS=7 python tools/gen_fwd_bwd_bench.py 32
TL=7 prof cfg1_b4_bf16 $PB --precision bf16_data
# PMC passes (each counter set in its own run, kernel-trace only)
bash tools/pmc_bench.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_bench.json $O/round6_pmc.json 2>/dev/null
bash tools/pmc_northstar.sh > $O/pmc_ns.log 2>&1; cp gpurun_out/pmc_northstar.json $O/round6_pmc_northstar.json 2>/dev/null
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_ns
for f in $O/*.json; do echo "$f: $(cut -c1-140 $f)"; done
# the number the review recomputes: dominant fp32 family from the single-stream CSV
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/round6_kernel_stats_default_f32_single_stream.csv")))
print("single-stream CSV, top kernels:")
for r in rows[:6]:
    print("  ", {k: r[k] for k in list(r)[:6]})
PY
