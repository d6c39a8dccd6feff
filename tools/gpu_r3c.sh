#!/bin/bash
# full GPU suite on HEAD (incl. the new full-size / layer-chain / fused tests), the parity survey, then the bench line
mkdir -p gpurun_out/r3c
timeout -k 10 1100 python -m pytest tests/ -q -m gpu --durations=12 > gpurun_out/r3c/gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3c/gpu_tests.log
tail -60 gpurun_out/r3c/gpu_tests.log
timeout -k 10 600 python tools/model_parity_survey.py > gpurun_out/r3c/survey.log 2>&1
tail -40 gpurun_out/r3c/survey.log | cut -c1-420
timeout -k 10 400 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
tail -3 gpurun_out/r3c/bench.err; cut -c1-1200 gpurun_out/r3c/bench.json
