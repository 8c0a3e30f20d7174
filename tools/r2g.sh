R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r2g; cd /tmp
export SCAE_K8_FWD=${K8CFG:-2} SCAE_K8_DG=${K8CFG:-2} SCAE_K8_PAIR=${K8CFG:-2}
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r2g/x -o s -- python3 $R/tools/k8_pmc_one.py > $R/gpurun_out/r2g/x.log 2>&1
ls $R/gpurun_out/r2g/x
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/r2g/x/*counter_collection.csv")[0])))
print(rows[0].keys())
tr = {}
for f in glob.glob("$R/gpurun_out/r2g/x/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        tr[r['Dispatch_Id']] = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
agg = collections.defaultdict(dict)
for r in rows:
    agg[(r['Dispatch_Id'], r['Kernel_Name'][:90], r['Grid_Size'])][r['Counter_Name']] = float(r['Counter_Value'])
seen = set()
for (d, k, g), v in agg.items():
    if 'conv' not in k or (k, g) in seen: continue
    seen.add((k, g))
    us = tr.get(d, 0)
    print(k[28:90], g, "us %.1f" % us, {c: round(x) for c, x in v.items()}, "GRBM/us/8 = %.0f MHz" % (v.get('GRBM_GUI_ACTIVE',0)/max(us,1e-9)/8))
PY
rm -rf $R/gpurun_out/r2g/x
