#!/bin/bash
# rocprofv3 PMC passes (each in its own run, kernel-trace only) over the default bench workload -> gpurun_out/pmc_bench.json
#   gpurun -- bash tools/pmc_bench.sh        then copy to profiles/round1_pmc.json
set -e
OUT=$PWD/gpurun_out/pmc_bench
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star --no-config-legs --no-extra-legs --no-kernel-profile"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o f -- $CMD > $OUT/f.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o w -- $CMD > $OUT/w.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o m -- $CMD > $OUT/m.log 2>&1 || true
python - <<PY
import sqlite3, glob, json, re
res = {}
for f in sorted(glob.glob("$OUT/*results.db")):
    c = sqlite3.connect(f)
    for name, ctr, avg, cnt in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        res.setdefault(name, {})[ctr] = {"avg": avg, "launches": cnt}
TILES = {"128, 128": "128x128", "128, 64": "128x64", "64, 64": "64x64", "128, 32": "128x32", "32, 64": "32x64"}
out = {"note": "rocprofv3 --pmc, per-launch averages over the default bench workload (256x256, batch 4, fp32); FETCH/WRITE in KB as reported", "kernels": {}}
for name, d in res.items():
    m = re.match(r"void pg::(conv|wgrad)_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, (\d+))?(?:, (\d+))?>", name)
    if not m:
        continue
    if m.group(1) == "conv":
        key = "conv_igemm<%sx%s,A%s,B%s>" % (m.group(2), m.group(3), m.group(6), m.group(7))
    else:
        key = "wgrad_igemm<%sx%s,xs%s,ys%s>" % (m.group(2), m.group(3), m.group(7), m.group(8))
        if m.group(9) is not None:          # compile-time geometry / mask variants: x_is_large, has-mask
            key = key[:-1] + ",xl%s,hm%s>" % (m.group(9), m.group(10))
    e = {"launches": max(v["launches"] for v in d.values())}
    for ctr, v in d.items():
        e[ctr + ("_KB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else "")] = v["avg"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
        # 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)
    if key not in out["kernels"] or out["kernels"][key]["launches"] < e["launches"]:    # template variants of one family
        out["kernels"][key] = e
json.dump(out, open("$PWD/gpurun_out/pmc_bench.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["launches"])[:6]:
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
