#!/bin/bash
# rocprofv3 passes over the single-GPU path where it is HBM-bound (2 M = config 4's per-rank size, 16 M = config 4's
# whole scene): kernel-trace stats, then FETCH_SIZE / WRITE_SIZE / wait counters in passes of their own.
# Run on the GPU box from the repo root; results under gpurun_out/hbm_<n>_<pass>/.  tools/refresh_profiles.sh condenses them.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for n in ${SIZES:-2000000 16000000}; do
    rm -rf $O/hbm_${n}_*
    timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/hbm_${n}_stats -o kt -- python3 $R/tools/path_only.py 10 $n > $O/hbm_${n}_stats.log 2>&1 \
        || { tail -20 $O/hbm_${n}_stats.log; exit 1; }
    for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS"; do
        tag=${c%% *}
        timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d $O/hbm_${n}_$tag -o pmc -- python3 $R/tools/path_only.py 6 $n > $O/hbm_${n}_$tag.log 2>&1 \
            || { tail -20 $O/hbm_${n}_$tag.log; exit 1; }
    done
    echo "n=$n ok"
done
