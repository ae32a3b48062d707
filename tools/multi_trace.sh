#!/bin/bash
# Kernel + memcpy timeline of the one-rank protocol rehearsal (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/multi_trace
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/multi_trace -o mt -- python3 $R/tools/multi_one_rank.py > $O/multi_trace.log 2>&1
tail -4 $O/multi_trace.log
