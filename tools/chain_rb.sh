#!/bin/bash
# capsule-MLP chain (K7b): 16 vs 32 batch rows per workgroup at the cfg-3 shape, in the step
out=${1:-gpurun_out/chain_rb}; mkdir -p $out
for rb in 16 32; do
  SCAE_CHAIN_RB=$rb python bench.py --workload mnist_48_64_bs1024 --steps 20 --warmup 5 --blocks 5 --no-extra --no-cpu-baseline --no-roofline > $out/bench_rb$rb.json 2> $out/err_rb$rb.txt
  python tools/bench_brief.py $out/bench_rb$rb.json | head -1
  ( export TMPDIR=/tmp SCAE_CHAIN_RB=$rb; R=$PWD; cd /tmp; rm -rf /tmp/ch_$rb
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ch_$rb -o s -- python3 $R/bench.py --workload mnist_48_64_bs1024 --steps 10 --warmup 3 --blocks 1 --no-extra --no-cpu-baseline --no-roofline > /dev/null 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/ch_$rb/**/s_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if any(k in r['Name'] for k in ('chain_kernel', 'gemm_multi', 'bwd_cell', 'conv_bwd')):
            print("   rb $rb", r['Name'][:70], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1))
PY
  )
done
