#!/bin/bash
# usage: tools/quick_gpu.sh <tag>  -- gpu tests + short bench + step segmentation
R=$PWD; mkdir -p $R/gpurun_out/$1; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $R/gpurun_out/$1/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/$1/bench.json 2> $R/gpurun_out/$1/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$1 -o step -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/$1/bench_prof.log 2>&1)
python3 $R/tools/step_seg.py $R/gpurun_out/$1/step_kernel_trace.csv full > $R/gpurun_out/$1/seg.txt 2>&1
rm -f $R/gpurun_out/$1/step_kernel_trace.csv
tail -5 $R/gpurun_out/$1/pytest_gpu.txt; head -c 250 $R/gpurun_out/$1/bench.json; echo; head -1 $R/gpurun_out/$1/seg.txt
