#!/bin/bash
# usage: tools/lanes_prof.sh <tag> <graph|one|two> -- one replayed step as a timeline
R=$PWD; T=$1; M=$2; O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o $M -- python3 $R/tools/lanes_run.py $M 30 > $O/prof_$M.log 2>&1)
python3 $R/tools/step_timeline.py $O/${M}_kernel_trace.csv > $O/timeline_$M.txt 2>&1
rm -f $O/${M}_kernel_trace.csv $O/${M}_agent_info.csv
cat $O/timeline_$M.txt
