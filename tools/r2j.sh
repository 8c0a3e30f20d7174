R=$PWD; T=r2j; mkdir -p $R/gpurun_out/$T; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -k "trunk_on_matrix or full_size or fused_set or flat_gradient or trajectory" 2>&1 | tail -30 > $R/gpurun_out/$T/pytest_gpu.txt
grep -v "^    \|^$" $R/gpurun_out/$T/pytest_gpu.txt | tail -25
