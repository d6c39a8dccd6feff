#!/bin/bash
# as3 kernel: parity, stamps, then same-box A/B
mkdir -p gpurun_out/r3i
timeout -k 10 600 python -m pytest tests/test_gemm_split_gpu.py tests/test_gemm_ws_gpu.py -x -q > gpurun_out/r3i/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3i/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 ./tools/ubench/gemm_as3_stamp > gpurun_out/r3i/stamp.log 2>&1 && timeout -k 10 120 ./tools/ubench/gemm_as3_stamp 327680 >> gpurun_out/r3i/stamp.log 2>&1; cat gpurun_out/r3i/stamp.log
timeout -k 10 300 python tools/ab_bench.py --reps 3 base noas3 > gpurun_out/r3i/ab.log 2>&1 && tail -4 gpurun_out/r3i/ab.log
