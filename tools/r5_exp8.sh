#!/bin/bash
# round 5: persistent tap-pair kernel, second version (lane constants re-derived per tile) — per-variant masks
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
PG_PAIR_PERSIST=31 timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kernels.py -q -m gpu -k "x_phase or conv_bf16_big_kernel" > gpurun_out/r5/pp2_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5/pp2_tests.log | cut -c1-200
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5/exp8.txt -- \
  "ns off|PG_PAIR_PERSIST=0|$NS" "ns 128s|PG_PAIR_PERSIST=3|$NS" "ns 128s+256m|PG_PAIR_PERSIST=11|$NS" "ns all256|PG_PAIR_PERSIST=15|$NS" "ns all|PG_PAIR_PERSIST=31|$NS" \
  "ns off|PG_PAIR_PERSIST=0|$NS" "ns 128s|PG_PAIR_PERSIST=3|$NS" "ns 128s+256m|PG_PAIR_PERSIST=11|$NS" "ns all256|PG_PAIR_PERSIST=15|$NS" "ns 256 only|PG_PAIR_PERSIST=4|$NS"
echo "== layer bench, persist 0 / 15"; for v in 0 15; do PG_PAIR_PERSIST=$v python tools/layer_bench.py 32 enc1 enc2 enc3 dec3 dec4 dec5 2>/dev/null | grep -v amdgpu; done
