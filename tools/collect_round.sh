#!/bin/bash
# gpurun_out/<TAG>prof[_f32] (tools/prof_round.sh) -> profiles/<TAG>_pmc[_f32]/ + profiles/<TAG>_bench_kernel_stats[_f32].csv
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}; SUF=""; [ "${2:-}" = "f32" ] && SUF="_f32"
src=gpurun_out/${TAG}prof$SUF
mkdir -p profiles/${TAG}_pmc$SUF
for p in p1 p2 p3 p4 p5 p6; do cp "$(ls -t $src/pmc/$p/*/*_counter_collection.csv | head -1)" profiles/${TAG}_pmc$SUF/${p}_counter_collection.csv; done
cp $src/pmc_summary.json profiles/${TAG}_pmc$SUF/summary.json
cp "$(ls -t $src/stats/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_bench_kernel_stats$SUF.csv
ls -la profiles | grep ${TAG}_
