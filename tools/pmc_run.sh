#!/bin/bash
# Counter passes over the unmodified bench (fused stream kernel) and the stage-1-only probe.
# usage: tools/pmc_run.sh <tag>   (run on the GPU box; one rocprofv3 run per pass)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR"
timeout -s KILL 150 rocprofv3 --pmc $A --output-format csv -d $R/gpurun_out/pmc_$1_benchA -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$1_benchA.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc $B --output-format csv -d $R/gpurun_out/pmc_$1_benchB -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$1_benchB.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc $A --output-format csv -d $R/gpurun_out/pmc_$1_s1A -o p -- python3 $R/tools/exp_stream.py 16 > $R/gpurun_out/pmc_$1_s1A.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc $B --output-format csv -d $R/gpurun_out/pmc_$1_s1B -o p -- python3 $R/tools/exp_stream.py 16 > $R/gpurun_out/pmc_$1_s1B.log 2>&1
