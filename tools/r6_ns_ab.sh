#!/bin/bash
# round 6: north-star pass (generator forward + backward, batch 32, bf16 data path) A/B: quad kernels off / 4-wave / 8-wave
mkdir -p gpurun_out
rm -f gpurun_out/r6_ns_ab.log
for cfg in "PG_BIG_QUAD=0" "PG_BIG_QUAD=1 PG_QUAD_WAVES=4" "PG_BIG_QUAD=1 PG_QUAD_WAVES=8" "PG_BIG_QUAD=0" "PG_BIG_QUAD=1 PG_QUAD_WAVES=4"; do
  echo "---- $cfg" >> gpurun_out/r6_ns_ab.log
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-config-legs --no-extra-legs --no-kernel-profile --steps 5 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ns = d['north_star']
print('north_star ms %.3f frac %.4f | b32 step %s img/s | fp32 b4 %.1f img/s' % (ns['ms'], ns['frac_of_bf16_peak'], d['bf16_data_b32_img_s']['value'], d['value']))
for k, v in ns['families'].items(): print('   %-34s n=%2d %.3f ms %.0f TF' % (k, v['launches'], v['ms'], v['tflops']))
" >> gpurun_out/r6_ns_ab.log 2>&1
done
cat gpurun_out/r6_ns_ab.log
