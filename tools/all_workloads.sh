#!/bin/bash
for w in mnist_24_24_bs128 mnist_40_32_bs128 cifar_32_32_bs256 mnist_48_64_bs1024; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>&1 | tail -1 | cut -c1-420
done
