#!/bin/bash
# After `tools/final_gpu_run.sh tests`, `... profiles` and `... profiles2` on the GPU box: copy the summaries the box
# condensed (gpurun_out/summary/) into the committed profiles/ files of round $ROUND (default r04).
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r04}
S=gpurun_out/summary
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-pmc   (MI355X, default K=50 W=10; summarised by"
  echo "# tools/summarize_prof.py from the rocpd results db).  k_scatter<unsigned int, 4, 16, 512, false> aggregates every launch of"
  echo "# the 64 Mi-pair instance in the run: the roofline leg's pass-0 and pass-3 series (cold + warm-up + timed), and all four digit"
  echo "# passes of every whole sort (30-bit, uniform-32 and arange keys; arange pass 0 is the 0.4 ms outlier that lifts the average);"
  echo "# k_dbg_copy / k_dbg_tile_copy are the copy-ceiling leg (roofline.copy_ceiling).  bench.py's roofline = MEDIAN of 200 steady-state"
  echo "# pass-0 launches (one HIP event per launch): compare with median_us.  k_traverse / k_chunk rows: the 1 M path (config 2) plus"
  echo "# the config-3 (clustered, 25 M pairs: chunked allocation + k_pairs_compact), 2 M and 16 M steps and the reference's benchmark shapes."
  # the bench line this very run printed (same box, under the profiler), next to the one filed as ${R}_bench_final.json (another lease)
  python - <<PY
import json
try:
    mine = [json.loads(l) for l in open("gpurun_out/prof_bench.log") if l.startswith("{") and '"metric"' in l][-1]
    filed = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
    print("# THIS run's own bench line (the profiled process, this box): roofline launch_ms %.4f (frac %.4f), ms_per_step %.4f;  the line filed as"
          % (mine["roofline"]["launch_ms"], mine["roofline"]["frac"], mine["ms_per_step"]))
    print("# ${R}_bench_final.json comes from another lease (boxes differ by +- 6 %% on this kernel): launch_ms %.4f (frac %.4f), ms_per_step %.4f."
          % (filed["roofline"]["launch_ms"], filed["roofline"]["frac"], filed["ms_per_step"]))
except Exception as exc:
    print("# (no bench line of the profiled run: %r)" % (exc,))
PY
  cat $S/bench_kernel_stats.txt; } > profiles/${R}_bench_kernel_stats.txt
cp $S/radix64M_pmc.json profiles/${R}_radix64M_pmc.json
cp $S/path1M_pmc.json profiles/${R}_path1M_pmc.json
cp $S/path1M_float32_kernel_stats.txt profiles/${R}_path1M_kernel_stats.txt
cp $S/path1M_float64_kernel_stats.txt profiles/${R}_path1M_f64_kernel_stats.txt
for tag in 2M 16M; do
    cp $S/path${tag}_kernel_stats.txt profiles/${R}_path${tag}_kernel_stats.txt
    cp $S/path${tag}_pmc.json profiles/${R}_path${tag}_pmc.json
done
cp $S/traverse_leaf_blocks_pmc.json profiles/${R}_traverse_leaf_blocks_pmc.json
cp $S/config4_loopback_morton.json profiles/${R}_config4_loopback_morton.json
cp $S/config4_loopback_hash.json profiles/${R}_config4_loopback_hash.json
cp gpurun_out/bench_final.json profiles/${R}_bench_final.json
tail -1 gpurun_out/bench_g2.json > profiles/${R}_bench_gloo_n2_rehearsal.json
python - <<PY
import json
b = json.load(open("profiles/${R}_bench_final.json"))
print("value", b["value"], "ms", b["ms_per_step"], "frac", b["roofline"]["frac"], "launch_ms", b["roofline"]["launch_ms"],
      "traffic", b["roofline"]["traffic"], "Gkeys/s", b["radix_sort"]["gkeys_per_s"])
PY
sed -n 8,22p profiles/${R}_bench_kernel_stats.txt | cut -c1-118
