export PG_ONLY_BF16=1
run() { echo "== $*"; env "$@" python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"; }
run PG_WGTR4=1
run PG_WGTR4=0
run PG_WGTR4=2
run PG_WGTR4=1 PG_WGTR4_TARGET=128
run PG_WGTR4=2 PG_WGTR4_TARGET=128
run PG_WGTR4=1 PG_WGTR_TARGET=256
run PG_WGTR4=2 PG_WGTR_TARGET=256
run PG_WGTR4=1 PG_NO_SIDE_STREAM=1
run PG_WGTR4=2 PG_NO_SIDE_STREAM=1 PG_WGTR_TARGET=256
echo; PG_WGTR4=2 python tools/wgrad_bf16_bench.py
