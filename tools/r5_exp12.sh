#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for v in "PG_X=1" "PG_PAIR_STAGGER=6:1.0" "PG_PAIR_STAGGER=4:0.75" "PG_PAIR_STAGGER=8:0.5" "PG_PAIR_STAGGER=16:1.0"; do echo "== $v"; env $v python tools/layer_bench.py 32 enc1 enc2 dec3 dec4 dec5 2>/dev/null | grep -v amdgpu | cut -c1-90; done
NS="PG_ONLY_BF16=1 PG_NS_ITERS=30 python tools/gen_fwd_bwd_bench.py 32 | tail -1 | grep -o '[0-9.]* ms = [0-9.]* TFLOP/s'"
tools/r5_ab.sh gpurun_out/r5/exp12.txt -- "ns none|PG_X=1|$NS" "ns 6:1.0|PG_PAIR_STAGGER=6:1.0|$NS" "ns 8:0.5|PG_PAIR_STAGGER=8:0.5|$NS" "ns none|PG_X=1|$NS" "ns 6:1.0|PG_PAIR_STAGGER=6:1.0|$NS" "ns 16:1.0|PG_PAIR_STAGGER=16:1.0|$NS"
