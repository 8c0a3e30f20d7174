#!/bin/bash
# usage: tools/wl_prof.sh <tag> <workload> -- bench line + per-kernel totals of one step
R=$PWD; T=$1; WL=$2; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 900 python bench.py $EXTRA --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench_$WL.json 2> $R/gpurun_out/$T/bench_$WL.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T -o $WL -- python3 $R/bench.py $EXTRA --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/prof_$WL.log 2>&1
rm -f $R/gpurun_out/$T/${WL}_kernel_trace.csv
head -c 230 $R/gpurun_out/$T/bench_$WL.json; echo
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/$T/${WL}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}% calls {r["Calls"]:>5} avg_us {float(r["AverageNs"])/1e3:8.1f}  {r["Name"][:90]}')
PY
