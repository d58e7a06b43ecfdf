#!/bin/bash
# One profile set for profiles/: bench line, rocprofv3 kernel stats of the same command, PMC passes + summary.
# usage (on the GPU box, from the repo root):  bash tools/prof_all.sh r01_v7
set -u
TAG=${1:-r01_vX}
rm -rf gpurun_out/pmc gpurun_out/$TAG
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
cut -c1-400 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/$TAG.err || exit 1
cd $GRAFT_REPO_ROOT && head -5 gpurun_out/$TAG/bench_kernel_stats.csv | cut -c1-180
bash tools/pmc_passes.sh && python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.json && head -50 gpurun_out/pmc/summary.json
