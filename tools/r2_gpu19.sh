#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2
for SS in 0 1; do
  if [ $SS = 1 ]; then export PG_NO_SIDE_STREAM=1; else unset PG_NO_SIDE_STREAM; fi
  rm -rf $O/prof_ns$SS; mkdir -p $O/prof_ns$SS
  PG_ONLY_BF16=1 rocprofv3 --kernel-trace --stats -d $O/prof_ns$SS -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $O/prof_ns$SS/stdout.log 2>&1
  grep "generator fwd" $O/prof_ns$SS/stdout.log
  python tools/rocpd_summary.py $O/prof_ns$SS/ns_results.db $O/ns_stats_ss$SS.csv > /dev/null 2>&1
  rm -rf $O/prof_ns$SS
done
