#!/bin/bash
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r3s/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3s/tests.log
[ $rc -eq 0 ] || exit 1
cp gnnepcsaft_amd/libgnnepcsaft_hip.so /tmp/lib_new.so
for rep in 1 2 3; do
for which in new before; do
  if [ $which = before ]; then cp tools/ubench/lib_before.so gnnepcsaft_amd/libgnnepcsaft_hip.so; else cp /tmp/lib_new.so gnnepcsaft_amd/libgnnepcsaft_hip.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --launch eager > gpurun_out/r3s/bench_${which}.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r3s/bench_${which}.json')); print('$which rep$rep', round(d['ms_per_step'],3), [ (k['group'], round(k['ms_per_step_isolated'],3)) for k in d['kernels'] if 'batchnorm' in k['group']])"
done
done
cp /tmp/lib_new.so gnnepcsaft_amd/libgnnepcsaft_hip.so
