import os, sys, subprocess
for thr in ("4e9", "1e9", "2e8"):
    env = dict(os.environ, PG_WG_THR=thr)
    out = subprocess.run([sys.executable, "bench.py", "--precision", "bf16_data", "--no-cpu-baseline", "--no-kernel-profile"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    import json; j = json.loads(out); print(thr, j["value"], j["ms_per_step"])
