#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --timeout 1800 -x -q > gpurun_out/r2/tests_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -4 gpurun_out/r2/tests_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py ) 2>&1 | tail -5 | cut -c1-600
