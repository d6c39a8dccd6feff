#!/bin/bash
# the driver's round-end command on HEAD, then smoke()
mkdir -p gpurun_out/r3g
timeout -k 10 1100 python -m pytest tests/ -q -m gpu --durations=8 > gpurun_out/r3g/gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3g/gpu_tests.log
tail -25 gpurun_out/r3g/gpu_tests.log | cut -c1-300
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r3g/smoke.log 2>&1; tail -3 gpurun_out/r3g/smoke.log | cut -c1-400
