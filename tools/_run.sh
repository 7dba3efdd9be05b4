set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/t4.log 2>&1 || { tail -40 gpurun_out/r3/t4.log; exit 1; }
tail -2 gpurun_out/r3/t4.log
timeout -k 10 400 python bench.py > gpurun_out/r3/bench_full.log 2>&1 || { tail -20 gpurun_out/r3/bench_full.log; exit 1; }
tail -c 6000 gpurun_out/r3/bench_full.log
