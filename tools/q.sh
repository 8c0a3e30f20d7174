#!/bin/bash
# usage: tools/q.sh <tag> [pytest -k expr]  -- selected gpu tests + short bench (plain and forced rank-launcher)
R=$PWD; T=$1; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
if [ -n "$2" ]; then
  timeout 900 python -m pytest tests -q -m gpu -x -k "$2" 2>&1 | tail -25 > $R/gpurun_out/$T/pytest_gpu.txt
else
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $R/gpurun_out/$T/pytest_gpu.txt
fi
timeout 600 python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.json 2> $R/gpurun_out/$T/bench.err
tail -5 $R/gpurun_out/$T/pytest_gpu.txt; head -c 300 $R/gpurun_out/$T/bench.json; echo; tail -3 $R/gpurun_out/$T/bench.err
