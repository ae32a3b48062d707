#!/bin/bash
# After `tools/final_gpu_run.sh tests` and `... profiles` on the GPU box: condense gpurun_out/ into the committed
# profiles/ files of round $ROUND (default r02).
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r03}
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-pmc   (MI355X, default K=50 W=10; summarised by"
  echo "# tools/summarize_prof.py from the rocpd results db).  k_scatter<unsigned int, 4, 16, 512, false> aggregates every launch of"
  echo "# the 64 Mi-pair instance in the run: the roofline leg's pass-0 and pass-3 series (cold + warm-up + timed), and all four digit"
  echo "# passes of every whole sort (30-bit, uniform-32 and arange keys; arange pass 0 is the 0.4 ms outlier that lifts the average)."
  echo "# bench.py's roofline = MEDIAN of 200 steady-state pass-0 launches (one HIP event per launch): compare with median_us."
  echo "# k_traverse / k_chunk / k_scatter<.., 4, 4, 256, false> rows: the 1 M path (config 2) plus the config-3 (clustered, 25 M pairs)"
  echo "# and config-4-size (2 M) steps and the reference's benchmark shapes."
  python tools/summarize_prof.py stats gpurun_out/prof_bench/bench_results.db; } > profiles/${R}_bench_kernel_stats.txt
python tools/summarize_prof.py pmc gpurun_out/prof_pmc_FETCH_SIZE/pmc_results.db gpurun_out/prof_pmc_WRITE_SIZE/pmc_results.db \
    gpurun_out/prof_pmc_SQ_WAVE_CYCLES/pmc_results.db > profiles/${R}_radix64M_pmc.json
python tools/summarize_prof.py pmc gpurun_out/path_pmc_1/pmc_results.db gpurun_out/path_pmc_2/pmc_results.db gpurun_out/path_pmc_3/pmc_results.db \
    gpurun_out/path_pmc_4/pmc_results.db gpurun_out/path_pmc_5/pmc_results.db > profiles/${R}_path1M_pmc.json
cp gpurun_out/bench_final.json profiles/${R}_bench_final.json
tail -1 gpurun_out/bench_g2.json > profiles/${R}_bench_gloo_n2_rehearsal.json
# the HBM-bound regime: config 4's per-rank size (2 M) and its whole scene (16 M) on one GPU
for n in 2000000 16000000; do
    tag=$((n / 1000000))M
    { echo "# rocprofv3 --kernel-trace --stats -- python3 tools/path_only.py 10 $n   (uniform scene, contacts per sphere of config 2)"
      python tools/summarize_prof.py stats gpurun_out/hbm_${n}_stats/kt_results.db; } > profiles/${R}_path${tag}_kernel_stats.txt
    python tools/summarize_prof.py pmc gpurun_out/hbm_${n}_FETCH_SIZE/pmc_results.db gpurun_out/hbm_${n}_WRITE_SIZE/pmc_results.db \
        gpurun_out/hbm_${n}_SQ_WAVE_CYCLES/pmc_results.db > profiles/${R}_path${tag}_pmc.json
done
# the traversal before / after leaf blocks, uniform and clustered (tools/profile_traverse_r3.sh)
python - <<PY
import json, subprocess, sys
out = {}
for scene in ("uniform", "config3"):
    for tag in ("before", "after"):
        dbs = ["gpurun_out/trav_%s_%s_%d/pmc_results.db" % (scene, tag, i) for i in (1, 2)]
        d = json.loads(subprocess.check_output([sys.executable, "tools/summarize_prof.py", "pmc"] + dbs))
        out["%s_%s" % (scene, tag)] = {k: {c: v["median"] for c, v in cs.items()} for k, cs in d.items() if "k_traverse" in k}
json.dump(out, open("profiles/${R}_traverse_leaf_blocks_pmc.json", "w"), indent=1, sort_keys=True)
PY
tail -1 gpurun_out/config4_morton.log > profiles/${R}_config4_loopback_morton.json
tail -1 gpurun_out/config4_hash.log > profiles/${R}_config4_loopback_hash.json
python - <<PY
import json
b = json.load(open("profiles/${R}_bench_final.json"))
print("value", b["value"], "ms", b["ms_per_step"], "frac", b["roofline"]["frac"], "launch_ms", b["roofline"]["launch_ms"],
      "traffic", b["roofline"]["traffic"], "Gkeys/s", b["radix_sort"]["gkeys_per_s"])
PY
sed -n 8,20p profiles/${R}_bench_kernel_stats.txt | cut -c1-118
