#!/bin/bash
# rocprofv3 PMC passes (each counter group in its own run, kernel-trace only) over the north-star sub-metric workload
# (generator forward + backward, 256x256, batch 32, bf16 data path) -> gpurun_out/pmc_northstar.json
#   gpurun -- bash tools/pmc_northstar.sh      then copy to profiles/round2_pmc_northstar.json
OUT=$PWD/gpurun_out/pmc_ns
mkdir -p $OUT
export TMPDIR=/tmp PG_ONLY_BF16=1 PG_NS_ITERS=2 PG_NO_SIDE_STREAM=1
CMD="python tools/gen_fwd_bwd_bench.py 32"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o m -- $CMD > $OUT/m.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT -o l -- $CMD > $OUT/l.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o f -- $CMD > $OUT/f.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o w -- $CMD > $OUT/w.log 2>&1 || true
python - <<PY
import sqlite3, glob, json
res = {}
for f in sorted(glob.glob("$OUT/*results.db")):
    c = sqlite3.connect(f)
    try:
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print("no counters in", f, e); continue
    for name, ctr, avg, cnt in rows:
        res.setdefault(name, {})[ctr] = {"avg": avg, "launches": cnt}
out = {"note": "rocprofv3 --pmc (separate runs per counter group), per-launch averages; generator forward+backward 256x256 batch 32, "
               "bf16 data path, single stream; FETCH/WRITE in KB as reported (gfx950: double FETCH_SIZE for wide coalesced reads)", "kernels": {}}
for name, d in res.items():
    e = {"launches": max(v["launches"] for v in d.values())}
    for ctr, v in d.items():
        e[ctr] = v["avg"]
    if e.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)     # 1024 SIMDs; GUI_ACTIVE summed over 8 XCDs
    if e.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in e:
        e["lds_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][name[:110]] = e
json.dump(out, open("$PWD/gpurun_out/pmc_northstar.json", "w"), indent=1, sort_keys=True)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:10]
for k, v in top:
    print(k[:70], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("launches", "mfma_util", "lds_conflict_frac", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")})
PY
