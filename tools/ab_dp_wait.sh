#!/bin/bash
# one-GPU cost of the data-parallel reducer's stream ordering (single-rank communicator): side-stream events only vs events on both producer streams
export TMPDIR=/tmp
for E in "PG_FORCE_REDUCER=1" "PG_FORCE_REDUCER=1 PG_DP_WAIT_MAIN=1" "PG_FORCE_REDUCER=0"; do
  echo "== $E"
  env $E python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>&1 | tail -1 | cut -c60-125
  env $E python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-profile --precision bf16_data 2>&1 | tail -1 | cut -c60-125
done
python -m pytest tests -q -m gpu --timeout 900 -q -k "reducer or dp or rccl or full_size" 2>&1 | tail -2
