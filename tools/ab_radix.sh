#!/bin/bash
# A/B of two builds of the library on one box for the 64 Mi-pair sort: the in-tree build against
# tools/ab/libcollision_hip_prev.so (see tools/ab_builds.sh for how to make it), alternating processes.
#   bash tools/ab_radix.sh [rounds]
R=${1:-2}
for r in $(seq $R); do
    echo "== prev"; COLLISION_AMD_LIB=tools/ab/libcollision_hip_prev.so timeout -k 10 200 python tools/radix_ablate.py 4 0,2,16384 2>&1 | grep -v amdgpu.ids || exit 1
    echo "== new"; timeout -k 10 200 python tools/radix_ablate.py 4 0,2,16384 2>&1 | grep -v amdgpu.ids || exit 1
done
