#!/bin/bash
# Profiles of the default bench line (run on the GPU box): kernel trace + stats, then HBM read
# and write counters in separate passes.  Outputs land in gpurun_out/refresh/; tools/refresh_profiles.py
# (run afterwards, anywhere) copies the summaries into profiles/ and rewrites profiles/pmc_traffic.json.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
echo refreshed
