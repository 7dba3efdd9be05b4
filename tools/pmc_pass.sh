#!/bin/bash
# One rocprofv3 --pmc pass over a short default bench run (on the GPU box): tools/pmc_pass.sh <tag> <counters...> [-- bench args]
# Prints the mean per dispatch of every counter for the engine's kernels.
R=$GRAFT_REPO_ROOT
T=$1; shift
C=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do C+=("$1"); shift; done; [ "$1" = "--" ] && shift
mkdir -p $R/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc "${C[@]}" --output-format csv -d $R/gpurun_out/r3/pmc_$T -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --min-seconds 0 "$@" > $R/gpurun_out/r3/pmc_$T.log 2>&1 || { tail -5 $R/gpurun_out/r3/pmc_$T.log; exit 1; }
python3 - $R/gpurun_out/r3/pmc_$T <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "ssmap" not in k: continue
    acc[k.replace("void ", "").replace("ssmap::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:40].ljust(40), {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
