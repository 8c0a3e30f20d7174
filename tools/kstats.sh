#!/bin/bash
# usage: tools/kstats.sh <script.py> [args] -- per-kernel average durations (rocprofv3 --kernel-trace --stats)
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $R/"$@" > $O/log.txt 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/s_kernel_stats.csv")))[:14]:
    print(f'{float(r["AverageNs"])/1e3:8.2f} us x{r["Calls"]:>5}  {r["Name"][:100]}')
PY
rm -rf $O
