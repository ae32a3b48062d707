#!/bin/bash
# A/B of two builds of the library on one box: the in-tree build against tools/ab/libcollision_hip_prev.so (a build of an
# earlier commit, made with `git worktree` + make; *.so files travel with gpurun but stay out of the history), alternating
# processes.   bash tools/ab_builds.sh [rounds] [sizes...]
R=${1:-2}; shift
for r in $(seq $R); do
    COLLISION_AMD_LIB=tools/ab/libcollision_hip_prev.so timeout -k 10 200 python tools/path_time.py "$@" || exit 1
    timeout -k 10 200 python tools/path_time.py "$@" || exit 1
done
