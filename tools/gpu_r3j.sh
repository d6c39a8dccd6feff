#!/bin/bash
# ws3 predicate-free form: parity, then same-box A/B
mkdir -p gpurun_out/r3j
timeout -k 10 600 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_split_gpu.py -x -q > gpurun_out/r3j/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3j/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base nowsfast > gpurun_out/r3j/ab.log 2>&1 && tail -3 gpurun_out/r3j/ab.log; timeout -k 10 300 python -m pytest tests/test_fused_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 200 python tools/ws3_scaling.py > gpurun_out/r3j/ws3_scaling.log 2>&1; timeout -k 10 200 python tools/ws3_scaling.py nofast >> gpurun_out/r3j/ws3_scaling.log 2>&1; cat gpurun_out/r3j/ws3_scaling.log
