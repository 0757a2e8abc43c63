set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5_pytest1.log 2>&1; echo "pytest rc $?" 
tail -5 gpurun_out/r5_pytest1.log
B4="python bench.py --precision bf16_data --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 100 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
F4="python bench.py --no-cpu-baseline --no-north-star --no-config-legs --no-kernel-profile --steps 60 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])\""
NS="PG_ONLY_BF16=1 PG_NS_ITERS=20 python tools/gen_fwd_bwd_bench.py 32 | tail -1"
tools/r5_ab.sh gpurun_out/r5_exp1.txt -- \
  "ns par0|PG_ENC_PAR=0|$NS" "ns default|PG_X=1|$NS" "ns lvl5|PG_ENC_PAR_LEVEL=5|$NS" "ns lvl3|PG_ENC_PAR_LEVEL=3|$NS" "ns lvl1|PG_ENC_PAR_LEVEL=1|$NS" \
  "ns par0|PG_ENC_PAR=0|$NS" "ns default|PG_X=1|$NS" \
  "b4 par0|PG_ENC_PAR=0|$B4" "b4 default|PG_X=1|$B4" "b4 lvl2|PG_ENC_PAR_LEVEL=2|$B4" "b4 lvl0|PG_ENC_PAR_LEVEL=0|$B4" "b4 par0|PG_ENC_PAR=0|$B4" "b4 default|PG_X=1|$B4" \
  "f4 par0|PG_ENC_PAR=0|$F4" "f4 default|PG_X=1|$F4" "f4 lvl0|PG_ENC_PAR_LEVEL=0|$F4" "f4 par0|PG_ENC_PAR=0|$F4" "f4 default|PG_X=1|$F4"
