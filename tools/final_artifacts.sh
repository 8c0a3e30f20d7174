#!/bin/bash
# refresh the round's measurement artifacts (run on the GPU box from the repo root)
R=$PWD; mkdir -p $R/gpurun_out/final; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -2 > $R/gpurun_out/final/pytest_gpu.txt
# HBM traffic of the K8 kernels: separate --pmc passes (kernel trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/k8pmc_$c -o k8 -- python3 $R/tools/conv_pmc.py > /dev/null 2>&1)
done
python3 tools/pmc_k8.py > /dev/null
timeout 600 python bench.py > $R/gpurun_out/final/bench.json 2> $R/gpurun_out/final/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o bench -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/final/bench_under_rocprof.json 2> /dev/null)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o step -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > $R/gpurun_out/final/bench_step_only_under_rocprof.json 2> /dev/null)
python3 $R/tools/step_seg.py $R/gpurun_out/final/step_kernel_trace.csv full > $R/gpurun_out/final/step_sequence.txt 2>&1
rm -f $R/gpurun_out/final/*_kernel_trace.csv
cp $R/profiles/r01/k8_pmc.json $R/gpurun_out/final/k8_pmc.json
cat $R/gpurun_out/final/pytest_gpu.txt; tail -c 400 $R/gpurun_out/final/bench.json | head -c 300; echo; head -5 $R/gpurun_out/final/step_kernel_stats.csv | cut -c1-150
