#!/bin/bash
# SQ counter passes over the unmodified default bench (run on the GPU box): tools/pmc_bench.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"
timeout -s KILL 150 rocprofv3 --pmc $A --output-format csv -d $R/gpurun_out/pmc_$1_A -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$1_A.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc $B --output-format csv -d $R/gpurun_out/pmc_$1_B -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$1_B.log 2>&1
