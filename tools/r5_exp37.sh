#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/rw_patterns.hip -o /tmp/rw_patterns && /tmp/rw_patterns | tee gpurun_out/r5/exp37.txt
