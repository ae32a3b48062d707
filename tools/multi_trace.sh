#!/bin/bash
# Per-step GPU timeline of the one-rank protocol rehearsal (run on the GPU box from the repo root):
# rocprofv3 kernel trace of tools/multi_one_rank.py (Morton partition, every exchange forced), condensed by
# tools/multi_trace.py into the kernels of one steady-state step -> gpurun_out/multi_trace_step.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/multi_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/multi_trace -o t -- python3 $R/tools/multi_one_rank.py morton > $O/multi_trace.log 2>&1 \
    || { tail -20 $O/multi_trace.log; exit 1; }
cd $R
python tools/multi_trace.py $O/multi_trace > $O/multi_trace_step.txt && cat $O/multi_trace_step.txt
