#!/bin/bash
# Kernel-trace summaries of the other bench workloads (run on the GPU box after tools/refresh_profiles.sh);
# tools/refresh_profiles.py copies them into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
cd /tmp && export TMPDIR=/tmp
for w in goal_decay_c40 prob_hv_c40 w2_avg_prob_c150 avg_prob_c40 avg_prob_c21; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o p -- python3 $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $O/stats_$w.log 2>&1
  echo $w
done
