#!/bin/bash
mkdir -p gpurun_out/r3n
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_inference_gpu.py tests/test_fused_gpu.py tests/test_conv_gpu.py -x -q > gpurun_out/r3n/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3n/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/infer_bench.py > gpurun_out/r3n/infer.log 2>&1; grep "graphs" gpurun_out/r3n/infer.log
timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline > gpurun_out/r3n/bench_cfg1.json 2> gpurun_out/r3n/bench_cfg1.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3n/bench_cfg1.json'))
print(d['ms_per_step'], d['value'], d['config'].get('launch'), d['config'].get('launch_autotune'))
PY
