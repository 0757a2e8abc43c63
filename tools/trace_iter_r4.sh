# rocprofv3 kernel trace of the training iteration (default: 256x256, batch 4, bf16 data path) -> timeline text + census
#   gpurun -- bash tools/trace_iter_r4.sh [tag] [bench args...]
TAG=${1:-b4}; shift
ARGS=${@:---precision bf16_data --batch 4}
OUT=$PWD/gpurun_out/trace_it_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py $ARGS --steps 30 --warmup 10 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile > gpurun_out/bench_$TAG.json
rocprofv3 --kernel-trace -d $OUT -o it -- python bench.py $ARGS --steps 10 --warmup 5 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile > $OUT/stdout.log 2>&1 || true
python tools/timeline_r4.py $(ls $OUT/*results.db | head -1) gpurun_out/timeline_it_$TAG.txt 15
python tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) gpurun_out/kernel_stats_it_$TAG.csv
rm -rf $OUT
