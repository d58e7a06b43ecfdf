#!/bin/bash
# gpurun_out/r02c4 (tools/prof_config4.sh) -> profiles/<TAG>_config4_pmc/, <TAG>_config4_kernel_stats.csv, <TAG>_config4_roofline.json,
# and the bench command's roofline from profiles/<TAG>_pmc (tools/collect_round.sh first).  usage: bash tools/collect_config4.sh [TAG]
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r03}; src=gpurun_out/r02c4
mkdir -p profiles/${TAG}_config4_pmc
for p in p1 p2 p3 p4 p5 p6; do cp "$(ls -t $src/pmc/$p/*/*_counter_collection.csv | head -1)" profiles/${TAG}_config4_pmc/${p}_counter_collection.csv; done
cp $src/pmc_summary.json profiles/${TAG}_config4_pmc/summary.json
cp "$(ls -t $src/stats/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_config4_kernel_stats.csv
python3 tools/roofline_from_pmc.py profiles/${TAG}_config4_pmc/summary.json profiles/${TAG}_config4_kernel_stats.csv > profiles/${TAG}_config4_roofline.json
python3 tools/roofline_from_pmc.py profiles/${TAG}_pmc/summary.json profiles/${TAG}_bench_kernel_stats.csv > profiles/${TAG}_roofline.json
ls profiles | grep ${TAG}_config4
