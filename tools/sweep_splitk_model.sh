#!/bin/bash
# sweep of the split-K time-model constants (igemm_conv.hip) over the bench configurations
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
O=gpurun_out/r2
run() {  # tag, env..., -- bench args
  tag=$1; shift
  line=$(env "$@" 2>/dev/null | tail -1)
  python - "$tag" "$line" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    fam = d['roofline']['families']
    top = sorted(fam.items(), key=lambda kv: -kv[1]['ms'])[:3]
    print("%-34s %8.1f img/s %8.2f ms  " % (sys.argv[1], d['value'], d['ms_per_step']) + "  ".join("%s %.2fms %.0fTF" % (k[:28], v['ms'], v['tflops']) for k, v in top))
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
}
for F in 4 12 30; do
  for BW in 1.5 3.0; do
    E="PG_SPLITK_FIXED_US=$F PG_SPLITK_BW_TBS=$BW"
    run "f32 b4 F=$F BW=$BW" $E python bench.py --no-cpu-baseline --steps 10
    run "f32 cfg3 F=$F BW=$BW" $E python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --steps 10
    run "bf16 cfg3 F=$F BW=$BW" $E python bench.py --size 224 --pose_dim 32 --batch 8 --no-cpu-baseline --steps 10 --precision bf16_data
    run "bf16 b32 F=$F BW=$BW" $E python bench.py --batch 32 --no-cpu-baseline --steps 5 --warmup 2 --precision bf16_data
  done
done
