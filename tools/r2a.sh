tools/q.sh r2a "collective or trajectory"
timeout 600 python bench.py --gpus 1 --force-spawn --no-cpu-baseline --no-roofline > gpurun_out/r2a/bench_spawn.json 2> gpurun_out/r2a/bench_spawn.err
echo "spawn rc=$?"; head -c 900 gpurun_out/r2a/bench_spawn.json; tail -5 gpurun_out/r2a/bench_spawn.err
timeout 600 python bench.py --gpus 1 --force-spawn --no-overlap --no-cpu-baseline --no-roofline > gpurun_out/r2a/bench_spawn1.json 2> gpurun_out/r2a/bench_spawn1.err
echo "spawn1 rc=$?"; head -c 400 gpurun_out/r2a/bench_spawn1.json; tail -5 gpurun_out/r2a/bench_spawn1.err
SCAE_GRAPH_ALLREDUCE=1 timeout 600 python bench.py --gpus 1 --force-spawn --no-cpu-baseline --no-roofline > gpurun_out/r2a/bench_spawng.json 2> gpurun_out/r2a/bench_spawng.err
echo "spawn-ingraph rc=$?"; head -c 400 gpurun_out/r2a/bench_spawng.json; tail -5 gpurun_out/r2a/bench_spawng.err
