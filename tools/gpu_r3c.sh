#!/bin/bash
# full GPU suite on HEAD + the new full-size / layer-chain tests, then the bench line
mkdir -p gpurun_out/r3c
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r3c/gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3c/gpu_tests.log
tail -40 gpurun_out/r3c/gpu_tests.log
grep -q "rc=0" gpurun_out/r3c/gpu_tests.log || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
tail -3 gpurun_out/r3c/bench.err; cut -c1-1500 gpurun_out/r3c/bench.json
