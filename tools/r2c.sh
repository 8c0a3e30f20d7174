R=$PWD; T=r2c; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 -k "full_size or conv_stack_vs or mode_and_mean or recon_mse" 2>&1 | tail -40 > $R/gpurun_out/$T/pytest_gpu.txt
cat $R/gpurun_out/$T/pytest_gpu.txt
