#!/bin/bash
# rocprofv3 kernel statistics of the config-4 shard and a 2,000-scenario config-5 run (for profiles/r01_config{4,5}_kernel_stats.csv)
set -u
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c4prof -o c4 -- python3 $GRAFT_REPO_ROOT/tools/run_config4.py --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/c4prof.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5prof -o c5 -- python3 $GRAFT_REPO_ROOT/tools/run_config5.py --scenarios 2000 > $GRAFT_REPO_ROOT/gpurun_out/c5prof.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && head -4 gpurun_out/c4prof/c4_kernel_stats.csv | cut -c1-200 && head -4 gpurun_out/c5prof/c5_kernel_stats.csv | cut -c1-200
