#!/bin/bash
mkdir -p gpurun_out/r3p
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_fused_gpu.py -x -q > gpurun_out/r3p/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3p/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base rcloop > gpurun_out/r3p/ab.log 2>&1 && tail -3 gpurun_out/r3p/ab.log
timeout -k 10 300 python tools/ab_bench.py --config 5 --steps 8 --reps 2 base rcloop > gpurun_out/r3p/ab5.log 2>&1 && tail -3 gpurun_out/r3p/ab5.log
