# A/B of the north-star pass on ONE box: alternates the given environment settings, 30 timed passes each, 2 rounds
#   gpurun -- bash tools/ns_ab.sh "" "PG_NO_STEM_PF=1" "PG_NO_AUX_STREAM=1"
export PG_ONLY_BF16=1 PG_NS_ITERS=30
for round in 1 2; do
  for cfg in "$@"; do
    printf "%-40s " "[${cfg:-default}]"
    env $cfg python tools/gen_fwd_bwd_bench.py 32 2>/dev/null | grep -o "[0-9.]* ms = [0-9.]* TFLOP/s = [0-9.]*"
  done
done
