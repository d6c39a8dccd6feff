#!/bin/bash
# mid-size product kernel: parity, then the cfg-1 step
mkdir -p gpurun_out/r3l
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py tests/test_gemm_ws_gpu.py -x -q > gpurun_out/r3l/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3l/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline > gpurun_out/r3l/bench_cfg1.json 2> gpurun_out/r3l/bench_cfg1.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3l/bench_cfg1.json'))
print(d['ms_per_step'], d['value'], d['config'].get('launch'), d['config'].get('launch_autotune'))
for k in d.get('kernels',[]): print(k['group'], k['launches_per_step'], round(k['ms_per_step'],3), round(k.get('avg_us_isolated',0),1))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3l/stats -- python3 bench.py --config 1 --steps 20 --warmup 3 --no-cpu-baseline --launch eager > gpurun_out/r3l/bench_prof.json 2> gpurun_out/r3l/stats.err
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3l/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
