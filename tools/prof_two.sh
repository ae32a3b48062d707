set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
for n in 2000000 16000000; do
for v in 0 2097152; do
  rm -rf $O/pv_$v
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/pv_$v -o kt -- python3 $R/tools/path_only.py 20 $n auto uniform 3 $v > $O/pv_$v.log 2>&1 ) || { tail -5 $O/pv_$v.log; exit 1; }
  echo "== n $n variant $v"; python tools/summarize_prof.py stats $O/pv_$v/kt_results.db | cut -c1-120; rm -rf $O/pv_$v
done; done
