R=$PWD; T=r2r; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x -k "bf16" 2>&1 | tail -25 > $R/gpurun_out/$T/pytest_gpu.txt
grep -v "^    \|^$" $R/gpurun_out/$T/pytest_gpu.txt | tail -14
for a in "" "--bf16"; do timeout 600 python bench.py --workload mnist_48_64_bs1024 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $a | cut -c1-330; done
