tools/q.sh r2d "lazy_render or collective or trajectory or flat_gradient or golden or sum_rows"
R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2d -o plain -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2d/plain.log 2>&1
python3 $R/tools/step_timeline.py $R/gpurun_out/r2d/plain_kernel_trace.csv > $R/gpurun_out/r2d/timeline.txt 2>&1
rm -f $R/gpurun_out/r2d/plain_kernel_trace.csv
head -1 $R/gpurun_out/r2d/timeline.txt
