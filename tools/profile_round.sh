#!/bin/bash
# Collects the round's judged evidence on a GPU box into gpurun_out/<tag>/ (scratch); tools/collect_profiles.py copies
# the summaries into profiles/.  usage: bash tools/profile_round.sh r02
set -e
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT $OUT/stats $OUT/stats_ns $OUT/pmc/fetch $OUT/pmc/write $OUT/pmc/mfma $OUT/pmc/gui $OUT/stats_cfg3 $OUT/stats_cfg5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats/bench.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ns -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-stream > $OUT/stats_ns/bench.json 2> $OUT/stats_ns.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-stream > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-stream > /dev/null 2> $OUT/pmc_write.err
# matrix-pipe busy cycles and active cycles per dispatch (each counter in its own pass, kernel trace only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc/mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-stream --launch eager > /dev/null 2> $OUT/pmc_mfma.err || echo "mfma pmc failed"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/gui -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-stream --launch eager > /dev/null 2> $OUT/pmc_gui.err || echo "gui pmc failed"
for c in 1 3 4 5; do
  python3 bench.py --config $c --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err || echo "cfg $c failed"
done
for c in 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-side-stream > $OUT/stats_cfg$c/bench.json 2> $OUT/stats_cfg$c.err
done
echo done
