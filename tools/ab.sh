#!/bin/bash
# A/B timing of library builds on one box (run there): tools/ab.sh [-w workload] [-e envs] [-r rounds] name1 name2 ...
# Every name is semantic-search_amd/lib/ab/<name>.so ("cur" = the tree's own build), optionally followed
# by :VAR=value settings for that run (name:SSMAP_DIRECT=0); the builds take
# turns, `rounds` times, and every run prints frames/s, us per step and the stage times.
R=$GRAFT_REPO_ROOT
W=log_odds_c40; E=16; N=2
while getopts "w:e:r:" o; do case $o in w) W=$OPTARG;; e) E=$OPTARG;; r) N=$OPTARG;; esac; done
shift $((OPTIND-1))
for ((i=0;i<N;i++)); do
  for spec in "$@"; do
    n=${spec%%:*}; envs=""; [ "$spec" != "$n" ] && envs=$(echo "${spec#*:}" | tr ':' ' ')
    lib=$R/semantic-search_amd/lib/ab/$n.so; [ $n = cur ] && lib=$R/semantic-search_amd/lib/libssmap.so
    env $envs SSMAP_LIB=$lib timeout -k 10 150 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload $W --envs-per-gpu $E > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; exit 1; }
    python3 - "$spec" <<'PY'
import json,sys
for l in open('/tmp/ab.log'):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1].ljust(28), round(d['value']), round(d['ms_per_step']*1e3,1), {k[:6]: round(v*1e3,1) for k,v in d['stage_ms_per_step'].items()}, flush=True)
PY
  done
done
