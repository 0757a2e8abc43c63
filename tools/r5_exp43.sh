#!/bin/bash
# A/B of the warp-backward occupancy change at batch 4 on ONE box: current build, then the round-4 parameters rebuilt on the box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 200"
one() { $B4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
one new; one new
sed -i 's/#define PG_GATHER_PB 2/#define PG_GATHER_PB 4/; s/constexpr int U = 2;  /constexpr int U = 4;  /' pose-transfer_amd/csrc/warp.hip
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
one old; one old
