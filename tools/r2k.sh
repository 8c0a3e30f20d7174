R=$PWD; T=r2k; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -k "seed_attention or full_size or golden" 2>&1 | tail -30 > $R/gpurun_out/$T/pytest_gpu.txt
grep -v "^    \|^$" $R/gpurun_out/$T/pytest_gpu.txt | tail -12
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$T -o s -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.log 2>&1
python3 $R/tools/step_timeline.py $R/gpurun_out/$T/s_kernel_trace.csv > $R/gpurun_out/$T/timeline.txt 2>&1
rm -f $R/gpurun_out/$T/s_kernel_trace.csv
head -1 $R/gpurun_out/$T/timeline.txt; grep "stw_\|saw_\|sa_\|fold\|sum_rows" $R/gpurun_out/$T/timeline.txt
