#!/bin/bash
# A/B build of the library for kernel tuning: tools/build_variant.sh NAME [extra hipcc flags] -> ab/libmp_NAME.so
# (ab/ travels to the GPU box with gpurun; *.so is git-ignored).  Use with MP_HIP_LIB=ab/libmp_NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab
SCHED=""   # the product uses the default LLVM scheduler (__graft_entry__.SCHED_FLAGS)
if [ "${ILP_SCHED:-0}" = "1" ]; then SCHED="-mllvm -amdgpu-sched-strategy=iterative-ilp"; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value \
  $SCHED "$@" motion_planning_amd/csrc/mp_api.hip -o ab/libmp_$name.so
echo built ab/libmp_$name.so
