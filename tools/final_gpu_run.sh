#!/bin/bash
# Round-end GPU runs (from the repo root on the GPU box).  Two gpurun calls, each within the 1200 s limit:
#   bash tools/final_gpu_run.sh tests      GPU test suite, bench.py (default flags), N=2 gloo rehearsal of the bench
#   bash tools/final_gpu_run.sh profiles   rocprofv3 kernel-trace stats of bench.py + PMC passes (radix 64 Mi, 1 M path)
# tools/refresh_profiles.sh then condenses gpurun_out/ into the committed profiles/ files.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$1" = "tests" ]; then
    timeout -k 10 800 python -m pytest tests -m gpu -x -q --durations=10 > $O/final_tests.log 2>&1 || { tail -30 $O/final_tests.log; exit 1; }
    tail -14 $O/final_tests.log
    timeout -k 10 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -20 $O/bench_final.err; exit 1; }
    echo "bench ok"
    # typed as a plain command: bench.py starts its own torch.distributed.run (gloo here: both ranks share this box's GPU)
    COLLISION_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-radix > $O/bench_g2.json 2> $O/bench_g2.err \
        || { tail -20 $O/bench_g2.err; exit 1; }
    echo "gloo N=2 rehearsal ok"
    exit 0
fi
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_bench $O/prof_pmc_* $O/path_pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python3 $R/bench.py --no-cpu --no-pmc > $O/prof_bench.log 2>&1 \
    || { tail -20 $O/prof_bench.log; exit 1; }
echo "rocprof stats ok"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=${c%% *}
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/prof_pmc_$tag -o pmc -- python3 $R/tools/radix_only.py 2 > $O/prof_pmc_$tag.log 2>&1 \
        || { tail -20 $O/prof_pmc_$tag.log; exit 1; }
done
echo "radix pmc ok"
i=0
for c in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/path_pmc_$i -o pmc -- python3 $R/tools/path_only.py 10 > $O/path_pmc_$i.log 2>&1 \
        || { tail -20 $O/path_pmc_$i.log; exit 1; }
done
echo "path pmc ok"
bash $R/tools/profile_hbm_regime.sh || exit 1
bash $R/tools/profile_traverse_r3.sh || exit 1
cd $R
timeout -k 10 200 python tools/config4_loopback.py morton > $O/config4_morton.log 2>&1 || { tail -20 $O/config4_morton.log; exit 1; }
timeout -k 10 200 python tools/config4_loopback.py hash > $O/config4_hash.log 2>&1 || { tail -20 $O/config4_hash.log; exit 1; }
echo "config 4 loopback ok"
