#!/bin/bash
mkdir -p gpurun_out/r3u
timeout -k 10 600 python -m pytest tests/test_gemm_ws_gpu.py -x -q > gpurun_out/r3u/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3u/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base wshalf > gpurun_out/r3u/ab.log 2>&1 && tail -3 gpurun_out/r3u/ab.log
timeout -k 10 300 python tools/ab_bench.py --config 4 --steps 10 --reps 2 base wshalf > gpurun_out/r3u/ab4.log 2>&1 && tail -3 gpurun_out/r3u/ab4.log
