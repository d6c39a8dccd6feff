#!/bin/bash
mkdir -p gpurun_out/r3v
timeout -k 10 900 python -m pytest tests/test_fused_gpu.py tests/test_conv_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py -x -q > gpurun_out/r3v/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3v/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_bench.py --reps 3 base nosplitahead > gpurun_out/r3v/ab.log 2>&1 && tail -3 gpurun_out/r3v/ab.log
timeout -k 10 300 python tools/ab_bench.py --config 5 --steps 8 --reps 2 base nosplitahead > gpurun_out/r3v/ab5.log 2>&1 && tail -3 gpurun_out/r3v/ab5.log
timeout -k 10 200 python bench.py --no-cpu-baseline --launch graph > gpurun_out/r3v/bench_graph.json 2>gpurun_out/r3v/bench_graph.err; python -c "
import json; d=json.load(open('gpurun_out/r3v/bench_graph.json')); print('graph mode', d['ms_per_step'], d['config']['launch'])"
