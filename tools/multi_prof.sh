#!/bin/bash
# Per-kernel durations of the one-rank protocol step (tools/multi_one_rank.py morton) with the device-side owned count and
# with the host-side one (COLLISION_HOST_OWNED_COUNT=1): bash tools/multi_prof.sh   (on the GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp
for mode in dev host; do
    rm -rf $O/mprof_$mode
    ( cd /tmp && WARM=30 COLLISION_HOST_OWNED_COUNT=$([ $mode = host ] && echo 1) timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/mprof_$mode -o kt -- python3 $R/tools/multi_one_rank.py morton > $O/mprof_$mode.log 2>&1 ) || { tail -20 $O/mprof_$mode.log; exit 1; }
    python tools/summarize_prof.py stats $O/mprof_$mode/kt_results.db > $O/multi_one_rank_${mode}_kernel_stats.txt
    rm -rf $O/mprof_$mode
done
