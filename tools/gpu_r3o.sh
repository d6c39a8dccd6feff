#!/bin/bash
mkdir -p gpurun_out/r3o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3o/stats -- python3 tools/infer_single_trace.py 40 > gpurun_out/r3o/out.log 2> gpurun_out/r3o/err.log
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/r3o/stats/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernel us per call', tot/1e3/40, 'launches per call', calls/40)
for r in rows[:30]: print(r['Name'][:72].ljust(72), int(r['Calls'])/40, round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
