R=$PWD; T=r2i; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "set_encoder or fused or golden or seed" 2>&1 | tail -15 > $R/gpurun_out/$T/pytest_gpu.txt
tail -6 $R/gpurun_out/$T/pytest_gpu.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T -o s -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.log 2>&1
python3 $R/tools/step_timeline.py $R/gpurun_out/$T/s_kernel_trace.csv > $R/gpurun_out/$T/timeline.txt 2>&1
rm -f $R/gpurun_out/$T/s_kernel_trace.csv
head -1 $R/gpurun_out/$T/timeline.txt; grep "st_\|stw_\|sa_\|fold" $R/gpurun_out/$T/timeline.txt
