R=$PWD; T=r2e; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "conv_stack or part_encoder or conv" 2>&1 | tail -15 > $R/gpurun_out/$T/pytest_gpu.txt
tail -5 $R/gpurun_out/$T/pytest_gpu.txt
timeout 600 python tools/k8_time.py > $R/gpurun_out/$T/k8.txt 2>&1
cat $R/gpurun_out/$T/k8.txt | cut -c1-600
