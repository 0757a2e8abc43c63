mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r6_full_gpu_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r6_full_gpu_tests2.log
rm -f gpurun_out/r6_disc_storage_ab.log
B="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-extra-legs --no-kernel-profile --precision bf16_data"
for rep in 1 2; do for m in bf16 f32; do
  if [ $m = f32 ]; then export PG_DISC_F32_STORE=1; else unset PG_DISC_F32_STORE; fi
  echo "---- discriminator storage $m" >> gpurun_out/r6_disc_storage_ab.log
  $B --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256 b4', d['value'])" >> gpurun_out/r6_disc_storage_ab.log
  $B --batch 32 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256 b32', d['value'])" >> gpurun_out/r6_disc_storage_ab.log
  $B --size 224 --pose_dim 32 --batch 8 --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224 p32 b8', d['value'])" >> gpurun_out/r6_disc_storage_ab.log
  $B --size 512 --batch 8 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 b8', d['value'])" >> gpurun_out/r6_disc_storage_ab.log
done; done
unset PG_DISC_F32_STORE
grep -E "passed|failed|FAILED" gpurun_out/r6_full_gpu_tests2.log | tail -8; cat gpurun_out/r6_disc_storage_ab.log
