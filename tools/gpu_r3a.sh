#!/bin/bash
# first GPU call of round 3: the fused forward kernel's tests, then the same-box A/B of the step with / without it
mkdir -p gpurun_out/r3a
timeout -k 10 400 python -m pytest tests/test_fused_gpu.py -x -q > gpurun_out/r3a/fused_test.log 2>&1
echo "rc=$?" >> gpurun_out/r3a/fused_test.log
tail -15 gpurun_out/r3a/fused_test.log
grep -q "rc=0" gpurun_out/r3a/fused_test.log || exit 1
timeout -k 10 300 python tools/ab_bench.py --steps 30 --reps 3 base nofused > gpurun_out/r3a/ab_fused.log 2>&1
cat gpurun_out/r3a/ab_fused.log
