#!/bin/bash
# SQ counters of every kernel of the (eager) training step
R=$PWD; export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_step/$tag -o s -- python3 $R/bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_step/*/s_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r['Kernel_Name']); k = m.group(1) if m else 'other'
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    v = {c: sum(x)/len(x) for c, x in d.items()}
    wc = v.get('SQ_WAVE_CYCLES', 1)
    print(f"{k:28s} waves {v.get('SQ_WAVES',0):8.0f} wavecyc {wc:12.0f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} valu {v.get('SQ_INSTS_VALU',0):10.0f} lds {v.get('SQ_INSTS_LDS',0):9.0f} ldsidx {v.get('SQ_LDS_IDX_ACTIVE',0):10.0f} bankconf {v.get('SQ_LDS_BANK_CONFLICT',0):9.0f} salu {v.get('SQ_INSTS_SALU',0):9.0f} vmem_rd {v.get('SQ_INSTS_VMEM_RD',0):8.0f}")
PY
rm -rf $R/gpurun_out/pmc_step
