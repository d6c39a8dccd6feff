#!/bin/bash
# small-degree paths in edge_combine_bwd / GINE kernels: parity, then old-lib vs new-lib bench lines (same box)
mkdir -p gpurun_out/r3q
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_fused_gpu.py -x -q > gpurun_out/r3q/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3q/tests.log
[ $rc -eq 0 ] || exit 1
cp gnnepcsaft_amd/libgnnepcsaft_hip.so /tmp/lib_new.so
for rep in 1 2; do
for which in new before; do
  if [ $which = before ]; then cp tools/ubench/lib_before.so gnnepcsaft_amd/libgnnepcsaft_hip.so; else cp /tmp/lib_new.so gnnepcsaft_amd/libgnnepcsaft_hip.so; fi
  for c in 3 2; do
    timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 20 > gpurun_out/r3q/bench_${which}_cfg$c.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/r3q/bench_${which}_cfg$c.json')); print('$which cfg$c rep$rep', round(d['ms_per_step'],3), d['config'].get('launch'))"
  done
done
done
cp /tmp/lib_new.so gnnepcsaft_amd/libgnnepcsaft_hip.so
