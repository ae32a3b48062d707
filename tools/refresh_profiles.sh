#!/bin/bash
# After tools/final_gpu_run.sh on the GPU box: condense gpurun_out/ into the committed profiles/ files.
set -e
cd "$(dirname "$0")/.."
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu   (MI355X, default K=50 W=10; summarised by tools/summarize_prof.py"
  echo "# from the rocpd results db).  k_scatter<unsigned int, 4, 16, 512> aggregates every launch of the 64 Mi-pair instance in the"
  echo "# run: the roofline leg's pass-0 and pass-3 series (cold + warm-up + timed), and all four digit passes of every whole sort"
  echo "# (30-bit, uniform-32 and arange keys; arange pass 0 is the 0.4 ms outlier that lifts the average).  bench.py's roofline ="
  echo "# HIP-event average of 50 steady-state pass-0 launches: compare with median_us."
  echo "# k_traverse / k_chunk / k_scatter<.., 4, 4, 256> rows: the 1 M path (config 2) plus the 5 config-3 (clustered, 25 M pairs) steps."
  python tools/summarize_prof.py stats gpurun_out/prof_bench/bench_results.db; } > profiles/r01_bench_kernel_stats.txt
python tools/summarize_prof.py pmc gpurun_out/prof_pmc_FETCH_SIZE/pmc_results.db gpurun_out/prof_pmc_WRITE_SIZE/pmc_results.db \
    gpurun_out/prof_pmc_SQ_WAVE_CYCLES/pmc_results.db > profiles/r01_radix64M_pmc.json
cp gpurun_out/bench_final.json profiles/r01_bench_final.json
python - <<'PY'
import json
b = json.load(open("profiles/r01_bench_final.json"))
print("value", b["value"], "ms", b["ms_per_step"], "frac", b["roofline"]["frac"], "launch_ms", b["roofline"]["launch_ms"],
      "traffic", b["roofline"]["traffic"], "Gkeys/s", b["radix_sort"]["gkeys_per_s"])
PY
sed -n 7,9p profiles/r01_bench_kernel_stats.txt | cut -c1-110
