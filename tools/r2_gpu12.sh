#!/bin/bash
export TMPDIR=/tmp
python tools/host_overhead.py f32 2>&1 | tail -2
python tools/host_overhead.py bf16_data 2>&1 | tail -2
python bench.py --no-cpu-baseline --precision bf16_data 2>/dev/null | tail -1 | cut -c1-400
