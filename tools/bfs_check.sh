export PG_ONLY_BF16=1
python tools/gen_fwd_bwd_bench.py 32 2>&1 | tail -3
PG_NO_BF16_STORE=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_networks.py -x -q -m gpu -k "bf16" 2>&1 | tail -15
