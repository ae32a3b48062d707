# rocprofv3 kernel stats of the path: bash tools/prof_path.sh "<n> <plan> <scene>" ...   (from the repo root on the GPU box)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  rm -rf $O/pp
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/pp -o kt -- python3 $R/tools/path_only.py 30 $1 ${2:-auto} ${3:-uniform} > $O/pp.log 2>&1 ) || { tail -5 $O/pp.log; exit 1; }
  echo "== $spec"; python tools/summarize_prof.py stats $O/pp/kt_results.db | cut -c1-120; rm -rf $O/pp
done
