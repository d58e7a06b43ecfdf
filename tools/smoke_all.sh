set -e
for cmd in "tools/run_scenario.py" "tools/run_scenario.py --method CEM-GRAD" "tools/run_sweep.py --trials 2" "tools/run_config4.py --steps 2" "tools/run_config5.py --scenarios 1000" "tools/cem_small.py" "tools/point_bench.py" "tools/latency_probe.py" "tests/manual/gpu_check.py"; do
  echo "=== $cmd"; timeout -k 10 280 python $cmd 2>&1 | grep -v amdgpu.ids | tail -6
done
