#!/bin/bash
# usage: tools/exp_bench_once.sh [bench args]   one bench run, value + stage times on one line
R=$GRAFT_REPO_ROOT
timeout -k 10 150 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['stage_ms_per_step'].items()})
"
