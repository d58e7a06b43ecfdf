#!/bin/bash
# Round-3 A/B on the GPU box: parity suite, then settled kernel times of the key shapes (fp64; cost+grad and cost only).
# usage (inside gpurun): bash tools/r03_ab.sh [tag]   -> gpurun_out/r03_ab_<tag>.txt
tag=${1:-main}
out=gpurun_out/r03_ab_$tag.txt
mkdir -p gpurun_out
{ timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread -k "${AB_K:-not plain_bench and not config4_job and not config5_job and not sweep_on_hip}" 2>&1 | tail -8; } > $out || { echo "TESTS FAILED"; cat $out; exit 1; }
MP_SHAPE_REPS=${AB_REPS:-80} timeout -k 10 300 python tools/shape_bench.py "${AB_SHAPES:-50,64,65536;13,101,65536;50,101,32768;13,64,65536;2000,256,8192}" >> $out 2>&1 || { echo "BENCH FAILED"; tail -5 $out; exit 1; }
grep -v amdgpu.ids $out
