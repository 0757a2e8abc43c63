for v in 448 256 192 128 64; do
  echo "== PG_BF16_BIG_MIN=$v"
  PG_BF16_BIG_MIN=$v python bench.py --no-cpu-baseline --no-kernel-profile --precision bf16_data 2>/dev/null | tail -1 | cut -c60-130
  PG_BF16_BIG_MIN=$v python bench.py --no-cpu-baseline --no-kernel-profile --precision bf16_data --size 224 --pose_dim 32 --batch 8 2>/dev/null | tail -1 | cut -c60-130
done
