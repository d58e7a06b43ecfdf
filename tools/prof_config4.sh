#!/bin/bash
# Round-2 profile of BASELINE config 4's per-GPU shard (2,000 boxes, 256 samples, 131,072 candidates, fp64 cost + gradient:
# split mode, clearance_kernel + eval_kernel<..., PRE>): kernel-trace stats and the PMC passes.  Outputs under
# gpurun_out/r02c4/, to be copied into profiles/r02_config4_*.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02c4
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/run_config4.py --steps 5 > $OUT/stats.log 2>&1 || { echo stats failed; tail -3 $OUT/stats.log; exit 1; }
PMC_OUT=$OUT/pmc PMC_PROG="tools/run_config4.py --steps 2" bash tools/pmc_passes.sh || exit 1
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.json || exit 1
ls $OUT $OUT/stats/*
