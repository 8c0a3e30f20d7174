R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r2b
cd /tmp
for mode in plain two one; do
  case $mode in
    plain) ARGS="";;
    two) ARGS="--force-spawn"; export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511;;
    one) ARGS="--force-spawn --no-overlap"; export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512;;
  esac
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2b -o $mode -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $ARGS > $R/gpurun_out/r2b/$mode.log 2>&1
  python3 $R/tools/step_timeline.py $R/gpurun_out/r2b/${mode}_kernel_trace.csv > $R/gpurun_out/r2b/$mode.timeline.txt 2>&1
  rm -f $R/gpurun_out/r2b/${mode}_kernel_trace.csv
  head -1 $R/gpurun_out/r2b/$mode.timeline.txt
done
