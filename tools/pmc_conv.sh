#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for one layer of tools/conv_bench.py.  gpurun -- bash tools/pmc_conv.sh dec4
set -e
LAYER=${1:-dec4}
OUT=$PWD/gpurun_out/pmc_$LAYER
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT -o p1 -- python tools/conv_bench.py 4 $LAYER > $OUT/p1.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o p2 -- python tools/conv_bench.py 4 $LAYER > $OUT/p2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o p3 -- python tools/conv_bench.py 4 $LAYER > $OUT/p3.log 2>&1 || true
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$OUT/*results.db")):
    c = sqlite3.connect(f)
    try:
        cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(f, "ERR", e); continue
    for r in rows:
        if "igemm" in r[0]:
            print(f.split("/")[-1], r[0][:60], r[1], "%.4g" % r[2], r[3])
PY
