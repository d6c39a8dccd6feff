#!/bin/bash
mkdir -p gpurun_out/r3b
for v in base v2; do timeout -k 10 120 ./tools/ubench/edge_fwd_stamp_$v > gpurun_out/r3b/stamp_$v.log 2>&1; cat gpurun_out/r3b/stamp_$v.log; done
timeout -k 10 400 python -m pytest tests/test_fused_gpu.py -x -q > gpurun_out/r3b/fused_test.log 2>&1
echo "rc=$?" >> gpurun_out/r3b/fused_test.log
tail -5 gpurun_out/r3b/fused_test.log
grep -q "rc=0" gpurun_out/r3b/fused_test.log || exit 1
timeout -k 10 300 python tools/ab_bench.py --steps 30 --reps 3 base nofused > gpurun_out/r3b/ab_fused.log 2>&1
cat gpurun_out/r3b/ab_fused.log
