#!/bin/bash
mkdir -p gpurun_out/r3h
timeout -k 10 500 python -m pytest tests/test_fused_gpu.py -x -q > gpurun_out/r3h/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3h/tests.log
tail -15 gpurun_out/r3h/tests.log | cut -c1-300
grep -q "rc=0" gpurun_out/r3h/tests.log || exit 1
timeout -k 10 300 python tools/ab_bench.py --steps 30 --reps 3 base nofusedbwd nofused > gpurun_out/r3h/ab.log 2>&1
cat gpurun_out/r3h/ab.log
timeout -k 10 500 python -m pytest tests/test_model_gpu.py tests/test_conv_gpu.py tests/test_trainer_gpu.py tests/test_full_size_gpu.py -q > gpurun_out/r3h/tests2.log 2>&1
echo "rc=$?" >> gpurun_out/r3h/tests2.log
tail -8 gpurun_out/r3h/tests2.log | cut -c1-300
