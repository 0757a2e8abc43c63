#!/bin/bash
# round 5 re-entry baseline: the whole GPU suite, smoke, the default driver line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 1800 -x > gpurun_out/r5/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -6 gpurun_out/r5/tests_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err ) 2>&1 | tail -4
tail -c 3000 gpurun_out/r5/bench_default.json
