#!/bin/bash
# usage (on the GPU box): tools/gpu_check.sh <tag> [tests]   GPU tests (optional), default bench, kernel trace
set -e
R=$GRAFT_REPO_ROOT
T=$1
mkdir -p $R/gpurun_out/r3
if [ "$2" = tests ]; then
  timeout -k 10 400 python -u -m pytest $R/tests -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/r3/${T}_tests.log 2>&1 || { tail -30 $R/gpurun_out/r3/${T}_tests.log; exit 1; }
  tail -2 $R/gpurun_out/r3/${T}_tests.log
fi
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/r3/${T}_bench.log 2>&1
python3 - <<PY
import json
for l in open("$R/gpurun_out/r3/${T}_bench.log"):
    if l.startswith('{'):
        d=json.loads(l); print("$T", round(d['value']), round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['stage_ms_per_step'].items()}, round(d['roofline']['frac'],3))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/${T}_kt -o p -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --min-seconds 0 > $R/gpurun_out/r3/${T}_kt.log 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/r3/${T}_kt 5
