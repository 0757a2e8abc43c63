#!/bin/bash
# rocprofv3 kernel statistics of the default bench workload with the weight gradients on the main stream (every kernel's
# duration is that of the kernel alone) -> gpurun_out/r2/default_ss.csv
export TMPDIR=/tmp PG_NO_SIDE_STREAM=1
O=gpurun_out/r2; mkdir -p $O/prof_ss
rocprofv3 --kernel-trace --stats -d $O/prof_ss -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile "$@" > $O/prof_ss/stdout.log 2>&1
tail -1 $O/prof_ss/stdout.log | cut -c1-200
python tools/rocpd_summary.py $O/prof_ss/p_results.db $O/default_ss.csv > /dev/null 2>&1
rm -rf $O/prof_ss
