#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for v in "PG_PAIR_PERSIST=0" "PG_PAIR_PERSIST=31 PG_PAIR_PERSIST_NOPF=1"; do echo "== $v"; env $v python tools/layer_bench.py 32 enc1 enc2 dec4 dec5 2>/dev/null | grep -v amdgpu | cut -c1-100; done
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
tools/r5_ab.sh gpurun_out/r5/exp30.txt -- "ns off|PG_PAIR_PERSIST=0|$NS" "ns walk all|PG_PAIR_PERSIST=31 PG_PAIR_PERSIST_NOPF=1|$NS" "ns walk 128s|PG_PAIR_PERSIST=3 PG_PAIR_PERSIST_NOPF=1|$NS" "ns off|PG_PAIR_PERSIST=0|$NS" "ns walk all|PG_PAIR_PERSIST=31 PG_PAIR_PERSIST_NOPF=1|$NS"
