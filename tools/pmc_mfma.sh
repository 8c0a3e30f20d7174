#!/bin/bash
# MFMA utilisation of the K8 / K7 kernels: one counter per pass (names vary between builds)
R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "mfma" | head -20
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_mfma/$c -o m -- python3 $R/tools/conv_pmc.py > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_mfma/*/m_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r['Kernel_Name']); k = m.group(1) if m else 'other'
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k, {c: round(sum(x)/len(x)) for c, x in sorted(d.items())}, "launches", max(len(x) for x in d.values()))
PY
rm -rf $R/gpurun_out/pmc_mfma
