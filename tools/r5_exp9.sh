#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
PG_PAIR_PERSIST=31 PG_PAIR_PERSIST_NOWAIT=1 timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kernels.py tests/test_gpu_round3.py -q -m gpu -k "x_phase or conv_bf16_big_kernel or north_star" > gpurun_out/r5/pp3_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5/pp3_tests.log | cut -c1-200
for v in "PG_PAIR_PERSIST=0" "PG_PAIR_PERSIST=15" "PG_PAIR_PERSIST=15 PG_PAIR_PERSIST_NOWAIT=1"; do echo "== $v"; env $v python tools/layer_bench.py 32 enc1 enc2 dec4 dec5 2>/dev/null | grep -v amdgpu; done
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5/exp9.txt -- "ns off|PG_PAIR_PERSIST=0|$NS" "ns all256 nowait|PG_PAIR_PERSIST=15 PG_PAIR_PERSIST_NOWAIT=1|$NS" "ns off|PG_PAIR_PERSIST=0|$NS" "ns all256 nowait|PG_PAIR_PERSIST=15 PG_PAIR_PERSIST_NOWAIT=1|$NS"
