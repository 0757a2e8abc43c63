#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/store_patterns.hip -o /tmp/store_patterns && /tmp/store_patterns | tee gpurun_out/r5/exp34.txt
