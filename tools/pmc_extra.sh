#!/bin/bash
# extra rocprofv3 counter passes (latency levels, instruction fetch, branch, LDS detail); outputs under gpurun_out/pmcx/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmcx}
PROG=${PMC_PROG:-tools/prof_run.py}
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name --pmc "$@" -- python3 $PROG > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; }
run x1 SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_LEVEL_WAVES && \
run x2 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES && \
run x3 SQ_BUSY_CYCLES SQ_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_BUSY_CU_CYCLES && echo all passes ok
python3 tools/pmc_summary.py $OUT
