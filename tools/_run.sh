set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3/t3.log 2>&1 || { tail -40 gpurun_out/r3/t3.log; exit 1; }
tail -2 gpurun_out/r3/t3.log
bash tools/ab.sh -r 2 base cur
bash tools/ab.sh -r 1 -w avg_prob_c40 base cur
bash tools/ab.sh -r 1 -w prob_hv_c40 base cur
bash tools/ab.sh -r 1 -w goal_decay_c40 base cur
bash tools/ab.sh -r 1 -w w2_avg_prob_c150 base cur
