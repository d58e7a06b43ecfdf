#!/bin/bash
# (gpurun merges every call's files into gpurun_out/: the newest file of each pass is the one taken.)
# Copy the summaries of tools/prof_r02.sh (gpurun_out/r02prof) and tools/prof_config4.sh (gpurun_out/r02c4) into profiles/.
set -e
cd "$(dirname "$0")/.."
if [ -d gpurun_out/r02prof ]; then
  src=gpurun_out/r02prof
  mkdir -p profiles/r02_pmc
  for p in p1 p2 p3 p4 p5 p6; do cp "$(ls -t $src/pmc/$p/*/*_counter_collection.csv | head -1)" profiles/r02_pmc/${p}_counter_collection.csv; done
  cp $src/pmc_summary.json profiles/r02_pmc/summary.json
  cp "$(ls -t $src/stats/*/*_kernel_stats.csv | head -1)" profiles/r02_bench_kernel_stats.csv
  cp $src/r02_workload_stats.json profiles/r02_workload_stats.json
  python3 - <<'PY'
import json
g = json.load(open("profiles/r02_pmc/summary.json"))["grad"]
json.dump({"eval_kernel_f64_bytes_per_launch": int((2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024),
           "source": "profiles/r02_pmc (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --profile-mode`; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction; KiB units)",
           "fetch_size_kib": g["FETCH_SIZE"], "write_size_kib": g["WRITE_SIZE"],
           "launch": "eval_kernel<double, 1, true, 0, false>, 65,536 candidates (config 3)",
           "breakdown": "algorithmic: 4.2 MB in (X) + 5.3 MB out (cost, gradient, min clearance, flags) = 9.5 MB.  Round 1 moved 142 MB "
                        "per launch: 109 MB of it were register-spill scratch writes, gone since no instance of the kernel uses scratch "
                        "memory any more (profiles/r02_kernel_resources.txt)"}, open("profiles/r02_traffic.json", "w"), indent=1)
PY
fi
if [ -d gpurun_out/r02c4 ]; then
  src=gpurun_out/r02c4
  mkdir -p profiles/r02_config4_pmc
  for p in p1 p2 p3 p4 p5 p6; do cp "$(ls -t $src/pmc/$p/*/*_counter_collection.csv | head -1)" profiles/r02_config4_pmc/${p}_counter_collection.csv; done
  cp $src/pmc_summary.json profiles/r02_config4_pmc/summary.json
  cp "$(ls -t $src/stats/*/*_kernel_stats.csv | head -1)" profiles/r02_config4_kernel_stats.csv
  grep '^{' $src/stats.log | tail -1 > profiles/r02_config4_shard.json || true
fi
python3 tools/kernel_resources.py > profiles/r02_kernel_resources.txt 2>/dev/null || true
ls -la profiles | grep r02
