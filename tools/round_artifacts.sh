#!/bin/bash
# usage: tools/round_artifacts.sh r02 [notests] -- refresh the round's measurement artifacts into
# gpurun_out/<round>/ (run on the GPU box from the repo root; copy what is to be judged into profiles/<round>/)
RN=${1:-r06}; R=$PWD; O=$R/gpurun_out/$RN; mkdir -p $O; export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|^E  |passed|failed" | tail -12 > $O/pytest_gpu.txt
fi
# HBM traffic of the K8 kernels: separate --pmc passes (kernel trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/k8pmc_$c -o k8 -- python3 $R/tools/conv_pmc.py > /dev/null 2>&1)
done
python3 tools/pmc_k8.py $O/k8_pmc.json > /dev/null
mkdir -p $R/profiles/$RN; cp $O/k8_pmc.json $R/profiles/$RN/k8_pmc.json   # bench.py reads traffic from here
# ... and of the K1 kernels; their SQ counters (VALU / LDS instruction counts, wait shares)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/k1pmc_$c -o k1 -- python3 $R/tools/k1_only.py > /dev/null 2>&1)
done
python3 tools/pmc_k1.py $O/k1_pmc.json > /dev/null
bash tools/pmc_k1.sh > $O/k1_sq_counters.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> /dev/null)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o step -- python3 $R/bench.py --steps 100 --warmup 10 --blocks 1 --no-cpu-baseline --no-roofline --no-extra > $O/bench_step_only_under_rocprof.json 2> /dev/null)
python3 $R/tools/step_timeline.py $O/step_kernel_trace.csv > $O/step_sequence.txt 2>&1
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
cat $O/pytest_gpu.txt; head -c 300 $O/bench.json; echo; head -3 $O/step_sequence.txt
# other workloads: bench line + per-kernel stats of one step (cfg-5, cfg-3 fp32 / bf16)
for spec in "cifar_32_32_bs256:" "mnist_48_64_bs1024:" "mnist_48_64_bs1024:--bf16"; do
  WL=${spec%%:*}; EX=${spec#*:}; TAG=$WL${EX:+_bf16}
  timeout 600 python bench.py $EX --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$TAG.json 2> /dev/null
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $TAG -- python3 $R/bench.py $EX --workload $WL --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2>&1)
  rm -f $O/${TAG}_kernel_trace.csv $O/${TAG}_agent_info.csv $O/${TAG}_domain_stats.csv
done
# the rank-launcher + RCCL path on one GPU (1-rank nccl group)
timeout 600 python bench.py --gpus 1 --force-spawn --steps 100 --no-cpu-baseline --no-roofline 2> /dev/null | grep '^{' > $O/bench_force_spawn.json
rm -f $O/*_domain_stats.csv
# upper bound of a multi-layer K8 launch and the image-resident forward (variant library built here)
[ -f tools/ablibs/libscae_multi.so ] && timeout 300 python tools/conv_multi_probe.py tools/ablibs/libscae_multi.so > $O/conv_multi_probe.txt 2>&1
[ -f tools/ablibs/libfwd_prof.so ] && timeout 300 python tools/fwd_prof.py tools/ablibs/libfwd_prof.so > $O/fwd_tile_timeline.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o gap -- python3 $R/tools/graph_gap_probe.py > /dev/null 2>&1)
python3 tools/graph_gap_report.py $O/gap_kernel_trace.csv > $O/graph_gap.txt 2>&1; rm -f $O/gap_*.csv
# round 6: do two plain streams overlap (yes); graph replay / launch list on one lane / on two lanes
[ -x tools/probes/stream_overlap ] && ./tools/probes/stream_overlap > $O/stream_overlap.txt 2>&1
RESIDENT=0,384,512,768 timeout 400 python tools/lanes_probe.py 2>&1 | grep -v amdgpu.ids > $O/lanes.txt
# the bf16-resident K8 kernels alone at cfg-3's layer shapes
timeout 200 python tools/conv_bf16_time.py 2>&1 | grep "B=1024" > $O/conv_bf16_time.txt
ls $O
