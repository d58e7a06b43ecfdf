#!/bin/bash
# Seed-grid A/B on the GPU box: parity suite, then settled kernel times of the key shapes with the grid off / on and at
# several resolutions (MP_GRID, MP_GRID_CELLS).  usage (inside gpurun): bash tools/grid_ab.sh   -> gpurun_out/grid_ab.txt
out=gpurun_out/grid_ab.txt
mkdir -p gpurun_out
SH="${AB_SHAPES:-50,64,65536;13,101,65536;50,101,32768;2000,256,8192;200,64,32768;2000,64,16384;50,256,16384}"
{ timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread -k "${AB_K:-not plain_bench and not config4_job and not config5_job and not sweep_on_hip}" 2>&1 | tail -8; } > $out || { echo "TESTS FAILED"; cat $out; exit 1; }
for cfg in "MP_GRID=0" "MP_GRID=1 MP_GRID_CELLS=1" "MP_GRID=1 MP_GRID_CELLS=4" "MP_GRID=1 MP_GRID_CELLS=16"; do
  echo "== $cfg" >> $out
  env $cfg MP_SHAPE_REPS=${AB_REPS:-60} timeout -k 10 300 python tools/shape_bench.py "$SH" >> $out 2>&1 || { echo "BENCH FAILED"; tail -5 $out; exit 1; }
done
grep -v amdgpu.ids $out
