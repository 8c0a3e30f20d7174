#!/bin/bash
# usage: tools/step_prof.sh <tag>  -- un-profiled bench line + one replayed step as a kernel timeline
R=$PWD; T=$1; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.json 2> $R/gpurun_out/$T/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$T -o s -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench_prof.log 2>&1
python3 $R/tools/step_timeline.py $R/gpurun_out/$T/s_kernel_trace.csv > $R/gpurun_out/$T/timeline.txt 2>&1
rm -f $R/gpurun_out/$T/s_kernel_trace.csv
head -c 260 $R/gpurun_out/$T/bench.json; echo; cat $R/gpurun_out/$T/timeline.txt | cut -c1-110
