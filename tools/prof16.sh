#!/bin/bash
# kernel-trace stats + FETCH / WRITE of the whole path at n spheres (default 16 M) -> gpurun_out/summary/prof_n_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; S=$O/summary; mkdir -p $S; cd $R
n=${1:-16000000}
export TMPDIR=/tmp
rm -rf $O/p16_stats
( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/p16_stats -o kt -- python3 $R/tools/path_only.py 10 $n > $O/p16_stats.log 2>&1 ) || { tail -20 $O/p16_stats.log; exit 1; }
python tools/summarize_prof.py stats $O/p16_stats/kt_results.db > $S/prof_${n}_stats.txt
dbs=""
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p16_$c
    ( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c -d $O/p16_$c -o pmc -- python3 $R/tools/path_only.py 6 $n > $O/p16_$c.log 2>&1 ) || { tail -20 $O/p16_$c.log; exit 1; }
    dbs="$dbs $O/p16_$c/pmc_results.db"
done
python tools/summarize_prof.py pmc $dbs > $S/prof_${n}_pmc.json
rm -rf $O/p16_*
head -14 $S/prof_${n}_stats.txt | cut -c1-130
python - <<PY
import json
d = json.load(open("$S/prof_${n}_pmc.json"))
for k, v in d.items():
    if any(t in k for t in ("k_chunk", "k_bucket_rows", "k_morton_tile", "k_traverse")):
        print(k[:60], {c: round(x["median"] / 1024, 1) for c, x in v.items()}, "MB (FETCH is half the bytes)")
PY
