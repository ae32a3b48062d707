#!/bin/bash
# PMC passes over the 1 M path (run on the GPU box from the repo root): bash tools/path_pmc.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for c in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf $O/path_pmc_$i
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/path_pmc_$i -o pmc -- python3 $R/tools/path_only.py 10 > $O/path_pmc_$i.log 2>&1 \
        || { tail -20 $O/path_pmc_$i.log; exit 1; }
done
echo "path pmc ok"
