#!/bin/bash
# A/B of two builds of the library on one box: the in-tree build against tools/ab/libcollision_hip_prev.so, alternating
# processes.  Make the other build first (it is not kept in the tree):
#   git worktree add /tmp/prev <commit> && make -C /tmp/prev/collision_amd/csrc && mkdir -p tools/ab &&
#   cp /tmp/prev/collision_amd/libcollision_hip.so tools/ab/libcollision_hip_prev.so && git worktree remove --force /tmp/prev
# (*.so files travel with gpurun but stay out of the history).   bash tools/ab_builds.sh [rounds] [sizes...]
R=${1:-2}; shift
for r in $(seq $R); do
    COLLISION_AMD_LIB=tools/ab/libcollision_hip_prev.so timeout -k 10 200 python tools/path_time.py "$@" || exit 1
    timeout -k 10 200 python tools/path_time.py "$@" || exit 1
done
