#!/bin/bash
# work counters of the clearance phase for the key shapes (needs ab/libmp_stats.so = -DMP_STATS build)
export MP_HIP_LIB=$PWD/${1:-ab/libmp_stats.so}
for shape in "50 64 16384" "13 101 16384" "2000 256 4096"; do timeout -k 10 120 python tools/clearance_stats.py $shape 2>&1 | grep -v amdgpu; done
