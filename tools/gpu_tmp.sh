#!/bin/bash
mkdir -p gpurun_out/r3x
timeout -k 10 600 python -m pytest tests/test_fused_gpu.py tests/test_trainer_gpu.py tests/test_dp_gpu.py -x -q > gpurun_out/r3x/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3x/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base notailearly notail > gpurun_out/r3x/ab.log 2>&1 && tail -4 gpurun_out/r3x/ab.log
timeout -k 10 300 python tools/ab_bench.py --config 4 --steps 10 --reps 2 base notailearly notail > gpurun_out/r3x/ab4.log 2>&1 && tail -4 gpurun_out/r3x/ab4.log
