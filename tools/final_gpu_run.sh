#!/bin/bash
# One gpurun call: GPU tests, bench, N=2 gloo rehearsal of the bench, rocprofv3 stats + PMC passes.
# Usage on the GPU box (from the repo root): bash tools/final_gpu_run.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/final_tests.log 2>&1 || { tail -30 $O/final_tests.log; exit 1; }
tail -2 $O/final_tests.log
timeout -k 10 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -20 $O/bench_final.err; exit 1; }
echo "bench ok"
COLLISION_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_g2.json 2> $O/bench_g2.err \
    || { tail -20 $O/bench_g2.err; exit 1; }
echo "gloo N=2 rehearsal ok"
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_bench $O/prof_pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python3 $R/bench.py --no-cpu > $O/prof_bench.log 2>&1 \
    || { tail -20 $O/prof_bench.log; exit 1; }
echo "rocprof stats ok"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=${c%% *}
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/prof_pmc_$tag -o pmc -- python3 $R/tools/radix_only.py 2 > $O/prof_pmc_$tag.log 2>&1 \
        || { tail -20 $O/prof_pmc_$tag.log; exit 1; }
done
echo "pmc ok"
find $O/prof_bench $O/prof_pmc_* -name "*.csv" | head -20
