#!/bin/bash
# PMC passes of the traversal before / after leaf blocks (round 3): uniform config 2 and clustered config 3, 1 M spheres.
# "before" = no marks (k = 0) and the walk without block code (variant 128); "after" = the production defaults.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for scene in uniform config3; do
  for tag in before after; do
    if [ $tag = before ]; then extra="0 128"; else extra=""; fi
    plan=auto; [ $scene = config3 ] && plan=lsd
    i=0
    for c in "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
      i=$((i+1))
      rm -rf $O/trav_${scene}_${tag}_$i
      timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d $O/trav_${scene}_${tag}_$i -o pmc -- python3 $R/tools/path_only.py 6 1000000 $plan $scene $extra > $O/trav_${scene}_${tag}_$i.log 2>&1 \
        || { tail -20 $O/trav_${scene}_${tag}_$i.log; exit 1; }
    done
  done
done
echo "traverse pmc ok"
