#!/bin/bash
# Round-2 profile set (GPU box): kernel-trace stats of the bench command, PMC passes over `bench.py --profile-mode`,
# workload statistics.  Outputs under gpurun_out/r02prof/, to be copied into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02prof
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --profile-mode --steps 200 --warmup 20 > $OUT/stats.log 2>&1 || { echo stats failed; tail -3 $OUT/stats.log; exit 1; }
PMC_OUT=$OUT/pmc PMC_PROG="bench.py --profile-mode --steps 4 --warmup 1" bash tools/pmc_passes.sh || exit 1
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.json || exit 1
MP_HIP_LIB=$PWD/ab/libmp_stats.so timeout -k 10 200 python3 tools/workload_stats.py > $OUT/workload_stats.log 2>&1 || { echo workload stats failed; tail -5 $OUT/workload_stats.log; exit 1; }
cp profiles/r02_workload_stats.json $OUT/
ls $OUT $OUT/stats/*
