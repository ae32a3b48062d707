#!/bin/bash
# Round-end GPU runs (from the repo root on the GPU box).  Each mode is one gpurun call within the 1200 s limit:
#   bash tools/final_gpu_run.sh tests       GPU test suite, bench.py (default flags), N=2 gloo rehearsal of the bench
#   bash tools/final_gpu_run.sh profiles    rocprofv3 kernel-trace stats of bench.py + PMC passes (radix 64 Mi, 1 M path)
#   bash tools/final_gpu_run.sh profiles2   the HBM-bound regime (2 M / 16 M), the traversal before / after leaf blocks, config 4
# The rocprofv3 databases are condensed ON THE BOX (tools/summarize_prof.py) into gpurun_out/summary/ and deleted: gpurun
# merges at most 64 MiB back.  tools/refresh_profiles.sh then copies the summaries into the committed profiles/ files.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
S=$O/summary
mkdir -p $S
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
pmc_pass() {      # pmc_pass <out dir> <counters> <script and args...>
    local d=$1 c=$2; shift 2
    rm -rf $d
    ( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python3 "$@" > $d.log 2>&1 ) || { tail -20 $d.log; exit 1; }
}
if [ "$1" = "profiles" ]; then
    rm -rf $O/prof_bench
    ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python3 $R/bench.py --no-cpu --no-pmc > $O/prof_bench.log 2>&1 ) \
        || { tail -20 $O/prof_bench.log; exit 1; }
    python tools/summarize_prof.py stats $O/prof_bench/bench_results.db > $S/bench_kernel_stats.txt && rm -rf $O/prof_bench
    echo "rocprof stats ok"
    dbs=""
    for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
        tag=${c%% *}
        pmc_pass $O/prof_pmc_$tag "$c" $R/tools/radix_only.py 2
        dbs="$dbs $O/prof_pmc_$tag/pmc_results.db"
    done
    python tools/summarize_prof.py pmc $dbs > $S/radix64M_pmc.json && rm -rf $O/prof_pmc_*
    echo "radix pmc ok"
    i=0; dbs=""
    for c in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM" \
             "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
             "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        pmc_pass $O/path_pmc_$i "$c" $R/tools/path_only.py 10
        dbs="$dbs $O/path_pmc_$i/pmc_results.db"
    done
    python tools/summarize_prof.py pmc $dbs > $S/path1M_pmc.json && rm -rf $O/path_pmc_*
    echo "path pmc ok"
    # per-kernel durations of the headline configuration alone (config 2, 1 M spheres), f32 and f64
    for dt in float32 float64; do
        rm -rf $O/path1M_$dt
        ( cd /tmp && COLLISION_PATH_DTYPE=$dt timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/path1M_$dt -o kt -- python3 $R/tools/path_only.py 50 1000000 > $O/path1M_$dt.log 2>&1 ) \
            || { tail -20 $O/path1M_$dt.log; exit 1; }
        { echo "# rocprofv3 --kernel-trace --stats -- python3 tools/path_only.py 50 1000000   (BASELINE config 2, $dt coordinates: 50 steps, every launch of the path)"
          python tools/summarize_prof.py stats $O/path1M_$dt/kt_results.db; } > $S/path1M_${dt}_kernel_stats.txt
        rm -rf $O/path1M_$dt
    done
    echo "path 1M kernel stats ok"
    exit 0
fi
# profiles2
for n in 2000000 16000000; do
    tag=$((n / 1000000))M
    rm -rf $O/hbm_${n}_stats
    ( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/hbm_${n}_stats -o kt -- python3 $R/tools/path_only.py 10 $n > $O/hbm_${n}_stats.log 2>&1 ) \
        || { tail -20 $O/hbm_${n}_stats.log; exit 1; }
    { echo "# rocprofv3 --kernel-trace --stats -- python3 tools/path_only.py 10 $n   (uniform scene, contacts per sphere of config 2)"
      python tools/summarize_prof.py stats $O/hbm_${n}_stats/kt_results.db; } > $S/path${tag}_kernel_stats.txt
    dbs=""
    for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS"; do
        t=${c%% *}
        pmc_pass $O/hbm_${n}_$t "$c" $R/tools/path_only.py 6 $n
        dbs="$dbs $O/hbm_${n}_$t/pmc_results.db"
    done
    python tools/summarize_prof.py pmc $dbs > $S/path${tag}_pmc.json && rm -rf $O/hbm_${n}_*
    echo "n=$n ok"
done
# the traversal before / after leaf blocks and chunked allocation (before: no marks, the walk without block code)
for scene in uniform config3; do
  for tag in before after; do
    extra=""; [ $tag = before ] && extra="0 128"
    plan=auto; [ $scene = config3 ] && plan=lsd
    i=0
    for c in "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
      i=$((i+1))
      pmc_pass $O/trav_${scene}_${tag}_$i "$c" $R/tools/path_only.py 6 1000000 $plan $scene $extra
    done
  done
done
python - <<PY
import json, subprocess, sys
out = {}
for scene in ("uniform", "config3"):
    for tag in ("before", "after"):
        dbs = ["$O/trav_%s_%s_%d/pmc_results.db" % (scene, tag, i) for i in (1, 2)]
        d = json.loads(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "pmc"] + dbs))
        out["%s_%s" % (scene, tag)] = {k: {c: v["median"] for c, v in cs.items()} for k, cs in d.items() if "k_traverse" in k or "k_pairs" in k}
json.dump(out, open("$S/traverse_leaf_blocks_pmc.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $O/trav_*
echo "traverse pmc ok"
timeout -k 10 200 python tools/config4_loopback.py morton > $O/config4_morton.log 2>&1 || { tail -20 $O/config4_morton.log; exit 1; }
timeout -k 10 200 python tools/config4_loopback.py hash > $O/config4_hash.log 2>&1 || { tail -20 $O/config4_hash.log; exit 1; }
tail -1 $O/config4_morton.log > $S/config4_loopback_morton.json
tail -1 $O/config4_hash.log > $S/config4_loopback_hash.json
echo "config 4 loopback ok"
