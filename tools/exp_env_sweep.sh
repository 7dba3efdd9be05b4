#!/bin/bash
# usage: tools/exp_env_sweep.sh VAR v1 v2 ...   runs the default bench once per value of an experiment variable
R=$GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 120 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['stage_ms_per_step'].items()})
" || exit 1
done
