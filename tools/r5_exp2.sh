set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PG_TOL_STUDY=1 timeout 1500 python -m pytest tests/test_gpu_round5.py -q -s -m gpu > gpurun_out/r5_pytest2.log 2>&1; echo "pytest rc $?"
grep -E "TOLSTUDY5|passed|failed|Error|assert" gpurun_out/r5_pytest2.log | tail -40
for i in 1 2; do
PG_DETERMINISTIC=1 PG_TRAJ_PRINT=1 timeout 900 python -m pytest tests/test_gpu_round4.py -q -s -m gpu -k trains_like > gpurun_out/r5_traj_det$i.log 2>&1; echo "traj rc $?"
grep -E "TRAJ|passed|failed" gpurun_out/r5_traj_det$i.log | cut -c1-400
done
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
B32="python bench.py --precision bf16_data --batch 32 --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 30 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NS="PG_ONLY_BF16=1 PG_NS_ITERS=20 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5_exp2.txt -- \
  "ns default(lvl1)|PG_X=1|$NS" "ns lvl0|PG_ENC_PAR_LEVEL=0|$NS" "ns default(lvl1)|PG_X=1|$NS" "ns lvl2|PG_ENC_PAR_LEVEL=2|$NS" \
  "b4 noeager|PG_NO_EAGER_ADAM=1|$B4" "b4 default|PG_X=1|$B4" "b4 noeager|PG_NO_EAGER_ADAM=1|$B4" "b4 default|PG_X=1|$B4" "b4 eager256k|PG_EAGER_ADAM_MIN=262144|$B4" "b4 eager8M|PG_EAGER_ADAM_MIN=8388608|$B4" \
  "f4 noeager|PG_NO_EAGER_ADAM=1|$F4" "f4 default|PG_X=1|$F4" "f4 noeager|PG_NO_EAGER_ADAM=1|$F4" "f4 default|PG_X=1|$F4" \
  "b32 noeager|PG_NO_EAGER_ADAM=1|$B32" "b32 default|PG_X=1|$B32" "b32 noeager|PG_NO_EAGER_ADAM=1|$B32" "b32 default|PG_X=1|$B32"
