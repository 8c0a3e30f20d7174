R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r2f; cd /tmp
export SCAE_K8_FWD=${K8CFG:-2} SCAE_K8_DG=${K8CFG:-2} SCAE_K8_PAIR=${K8CFG:-2}
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r2f/$tag -o s -- python3 $R/tools/k8_pmc_one.py > $R/gpurun_out/r2f/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/r2f/*/s_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:110] + " grid" + r.get('Grid_Size','?')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$R/gpurun_out/r2f/*/s_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:110] + " grid" + r.get('Grid_Size','?')
        dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, d in sorted(agg.items()):
    if "conv_fwd" not in k and "wgrad_pipe" not in k: continue
    v = {c: sum(x)/len(x) for c, x in d.items()}
    print(k)
    print("   dur_us %.1f" % (sum(dur[k])/max(1,len(dur[k]))), {c: round(x) for c, x in sorted(v.items())})
PY
rm -rf $R/gpurun_out/r2f/SQ_* $R/gpurun_out/r2f/GRBM*
