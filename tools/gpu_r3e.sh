#!/bin/bash
# fused forward kernel after the load re-ordering (stamps), fused / batched-weight-only tests, same-box A/Bs
mkdir -p gpurun_out/r3e

timeout -k 10 500 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_layer_chain_gpu.py tests/test_dropout_gpu.py -q > gpurun_out/r3e/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3e/tests.log
tail -15 gpurun_out/r3e/tests.log

timeout -k 10 300 python tools/ab_bench.py --steps 30 --reps 3 base nofused nobatchwo > gpurun_out/r3e/ab.log 2>&1
cat gpurun_out/r3e/ab.log
for cus in 128 160 192; do
  GNX_SIDE_CUS=$cus GNX_WGRAD_WGS=$cus timeout -k 10 200 python tools/ab_bench.py --steps 30 --reps 3 base > gpurun_out/r3e/ab_cus$cus.log 2>&1
  echo "side stream 0 on $cus CUs:"; cat gpurun_out/r3e/ab_cus$cus.log
done
timeout -k 10 200 python tools/ab_bench.py --config 1 --steps 50 --reps 3 base nobatchwo > gpurun_out/r3e/ab_cfg1.log 2>&1
cat gpurun_out/r3e/ab_cfg1.log
