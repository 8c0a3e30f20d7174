#!/bin/bash
# usage: tools/quick_timeline.sh <tag> [bench args] -- bench line + one-step timeline under rocprofv3
R=$PWD; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 200 "$@" > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o s -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline "$@" > $O/bench_prof.log 2>&1)
python3 $R/tools/step_timeline.py $O/s_kernel_trace.csv > $O/timeline.txt 2>&1
rm -f $O/s_kernel_trace.csv $O/s_agent_info.csv
head -c 260 $O/bench.json; echo; cat $O/timeline.txt
