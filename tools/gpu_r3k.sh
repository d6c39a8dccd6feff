#!/bin/bash
# 96-row tiles of the pipelined product: parity, then same-box A/B
mkdir -p gpurun_out/r3k
timeout -k 10 600 python -m pytest tests/test_gemm_split_gpu.py tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_fused_gpu.py -x -q > gpurun_out/r3k/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3k/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base rows128 rows96 > gpurun_out/r3k/ab.log 2>&1 && tail -4 gpurun_out/r3k/ab.log
timeout -k 10 300 python tools/ab_bench.py --config 4 --steps 10 --reps 2 base rows160 rows128 > gpurun_out/r3k/ab4.log 2>&1 && tail -4 gpurun_out/r3k/ab4.log
