#!/bin/bash
export TMPDIR=/tmp
for T in 768 512 256 1536; do
  echo "target $T"
  PG_WGTR_TARGET=$T PG_ONLY_BF16=1 python tools/gen_fwd_bwd_bench.py 32 2>&1 | grep "generator fwd"
  PG_WGTR_TARGET=$T python bench.py --precision bf16_data --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['roofline']['families'].items() if 'tr-256' in k})"
done
