# PMC counters of k_chunk at 1 M spheres for two col_debug_lbvh modes (from the repo root on the GPU box)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
for m in ${MODES:-8192 0}; do
  dbs=""
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
    i=$((i+1)); d=$O/pc_${m}_$i; rm -rf $d
    ( cd /tmp && COLLISION_LBVH_MODE=$m timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python3 $R/tools/path_only.py 10 ${N:-1000000} > $d.log 2>&1 ) || { tail -5 $d.log; exit 1; }
    dbs="$dbs $d/pmc_results.db"
  done
  echo "== mode $m"
  python tools/summarize_prof.py pmc $dbs | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'k_chunk' in k:
        print(k[:60]); print('  '+'  '.join('%s=%.0f'%(c.replace('SQ_',''),x['median']) for c,x in sorted(v.items())))
"
  rm -rf $O/pc_${m}_*
done
