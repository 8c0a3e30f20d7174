#!/bin/bash
# usage: tools/pmc_any.sh <kernel-name-substring> <script.py> [args] -- SQ counters of the matching kernels
R=$PWD; KEY=$1; shift; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_any/$tag -o s -- python3 $R/"$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_any/*/s_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$KEY" in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    for c, x in sorted(d.items()):
        print(f"   {c:32s} {sum(x)/len(x):14.0f}  (n={len(x)})")
PY
rm -rf $R/gpurun_out/pmc_any
