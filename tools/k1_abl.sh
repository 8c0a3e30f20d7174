#!/bin/bash
# K1 backward ablations: time + VALU instruction counts per build / form
out=${1:-gpurun_out/k1_abl}; R=$PWD
mkdir -p $out
run() {  # tag lib G NTB
  echo "== $1" >> $out/times.txt
  SCAE_HIP_LIB=$2 SCAE_GROUP_G=$3 SCAE_GROUP_NTB=$4 python tools/k1_time.py mnist_24_24_bs128 unit 2>/dev/null >> $out/times.txt
  ( export TMPDIR=/tmp SCAE_HIP_LIB=$R/$2 SCAE_GROUP_G=$3 SCAE_GROUP_NTB=$4; [ -z "$2" ] && unset SCAE_HIP_LIB; cd /tmp; rm -rf /tmp/pm_$1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d /tmp/pm_$1 -o s -- python3 $R/tools/k1_only.py > /dev/null 2>&1
    python3 - <<PY >> $R/$out/times.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm_$1/**/s_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bwd_" in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print("  ", k, {c: round(sum(x)/len(x)) for c, x in sorted(d.items())})
PY
  )
}
run new "" 0 0
run old tools/ablibs/libold.so 0 0
