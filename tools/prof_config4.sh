#!/bin/bash
# rocprofv3 kernel-trace stats of config 4 on one GPU (tools/config4_loopback.py: eight engines, 2 M spheres each):
# the kernels of a rank's step and their share -> gpurun_out/summary/config4_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; S=$O/summary; mkdir -p $S; cd $R
export TMPDIR=/tmp
rm -rf $O/c4_stats
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/c4_stats -o kt -- python3 $R/tools/config4_loopback.py ${1:-morton} > $O/c4_stats.log 2>&1 ) || { tail -20 $O/c4_stats.log; exit 1; }
{ echo "# rocprofv3 --kernel-trace --stats -- python3 tools/config4_loopback.py ${1:-morton}   (8 ranks x 2 M spheres in one process: 2 + 5 + 2 + 5 steps per rank; torch copy kernels = the loopback exchange)"
  python tools/summarize_prof.py stats $O/c4_stats/kt_results.db; } > $S/config4_kernel_stats.txt
rm -rf $O/c4_stats
head -40 $S/config4_kernel_stats.txt | cut -c1-140
