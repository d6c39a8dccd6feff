#!/bin/bash
mkdir -p gpurun_out/r3w
timeout -k 10 600 python -m pytest tests/test_gemm_ws_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r3w/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3w/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base embfp32 > gpurun_out/r3w/ab.log 2>&1 && tail -3 gpurun_out/r3w/ab.log
