#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_k1/$tag -o k1 -- python3 $R/tools/k1_only.py > $R/gpurun_out/pmc_k1_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_k1/*/k1_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r['Kernel_Name']); k = m.group(1) if m else 'other'
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'kernel' not in k: continue
    print(k, {c: round(sum(v)/len(v)) for c, v in sorted(d.items())})
PY
rm -rf $R/gpurun_out/pmc_k1/*/*.csv
