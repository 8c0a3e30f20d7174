#!/bin/bash
# per-launch durations of the GEMM / conv kernels of one cfg-3 step, fp32 vs bf16
R=$PWD; T=r2t; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp; cd /tmp
for mode in f32 bf16; do
  EXTRA=""; [ $mode = bf16 ] && EXTRA="--bf16"
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$T -o $mode -- python3 $R/bench.py $EXTRA --workload mnist_48_64_bs1024 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/$mode.log 2>&1
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/$T/${mode}_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stage_batch" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
out = []
for r in rows[a:b]:
    n = r["Kernel_Name"]
    if "gemm" in n or "conv" in n:
        out.append("%s %.0f" % (n.split("(")[0].split("::")[-1][:34], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("$mode", " | ".join(out))
PY
  rm -f $R/gpurun_out/$T/${mode}_kernel_trace.csv
done
