#!/bin/bash
# round 6: the whole GPU suite, then the prefetch gate A/B on the legs it changes (batch 32, 512^2)
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -q -m gpu -x ) > gpurun_out/r6_full_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r6_full_gpu_tests.log
rm -f gpurun_out/r6_prefetch_gate.log
B="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-extra-legs --no-kernel-profile --precision bf16_data"
for px in 786432 100000000 786432 100000000; do
  echo "---- PG_GEN_PREFETCH_MAX_PIX=$px" >> gpurun_out/r6_prefetch_gate.log
  PG_GEN_PREFETCH_MAX_PIX=$px $B --batch 32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32 bf16', d['value'])" >> gpurun_out/r6_prefetch_gate.log
  PG_GEN_PREFETCH_MAX_PIX=$px $B --size 512 --batch 8 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 b8 bf16', d['value'])" >> gpurun_out/r6_prefetch_gate.log
done
tail -15 gpurun_out/r6_full_gpu_tests.log; cat gpurun_out/r6_prefetch_gate.log
