#!/bin/bash
# Wider PMC sweep for one layer of tools/conv_bench.py (kernel-trace only, one counter group per pass).
LAYER=${1:-dec4}
OUT=$PWD/gpurun_out/pmc2_$LAYER
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TA_BUSY_avr TCP_GATE_EN1 TCP_GATE_EN2"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT -o p$i -- python tools/conv_bench.py 4 $LAYER > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$OUT/*results.db")):
    c = sqlite3.connect(f)
    try:
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(f, "ERR", e); continue
    for r in rows:
        if "conv_igemm" in r[0]:
            print(f.split("/")[-1][:12], r[0][24:60], r[1], "%.5g" % r[2], r[3])
PY
