set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wl
for w in log_odds_c40 avg_prob_c40 w2_avg_prob_c150 goal_decay_c40 prob_hv_c40 avg_prob_c21; do
  for E in 16 64; do
    if [ $w = avg_prob_c40 ] && [ $E = 64 ]; then continue; fi
    timeout -k 10 150 python bench.py --no-cpu-baseline --workload $w --envs-per-gpu $E --steps 100 > gpurun_out/wl/${w}_$E.log 2>&1
    echo $w $E
  done
done
