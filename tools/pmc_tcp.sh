#!/bin/bash
# L1 (TCP) behaviour of the config-3 launches: accesses, requests to L2, latency accumulators
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmct
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name --pmc "$@" -- python3 tools/prof_run.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }; }
run t1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum && \
run t2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum && echo ok
python3 tools/pmc_summary.py $OUT
