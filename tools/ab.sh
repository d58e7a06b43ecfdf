#!/bin/bash
# A/B on the GPU box: for each library variant (path or "main"), parity check + settled kernel times of the key shapes.
# usage (inside gpurun): bash tools/ab.sh main ab/libmp_x.so ...   -> gpurun_out/ab/<name>.txt
mkdir -p gpurun_out/ab
SHAPES=${AB_SHAPES:-"50,64,65536;13,101,65536;50,101,32768;200,64,32768;2000,256,8192"}
for v in "$@"; do
  name=$(basename "$v" .so)
  if [ "$v" = "main" ]; then unset MP_HIP_LIB; else export MP_HIP_LIB=$PWD/$v; fi
  { timeout -k 10 120 python tests/manual/quick_bench.py && MP_SHAPE_REPS=${AB_REPS:-60} timeout -k 10 200 python tools/shape_bench.py "$SHAPES"; } > gpurun_out/ab/$name.txt 2>&1 || { echo "variant $name FAILED"; tail -5 gpurun_out/ab/$name.txt; exit 1; }
  echo "== $name"; grep -v "amdgpu.ids" gpurun_out/ab/$name.txt
done
