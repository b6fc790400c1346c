#!/bin/bash
# L2 / vector-cache counters of the march kernel on the headline bench (run on the GPU box through gpurun); separate --pmc passes.
# usage: bash tools/pmc_l2.sh [tag]     (LES_HIP_LIB / LES_HIP_STATS_ROWPAD are honoured)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-l2}
O=gpurun_out/pmc_$TAG; mkdir -p $O
B="python bench.py --steps 3 --warmup 1 --cpu-planes 0 --sub-steps 0"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/p$i -- $B > $O/p$i.log 2>&1
  python tools/prof_summary.py $O/p$i les_march --md 2>/dev/null | grep -v "^##\|^|---\|^| counter\|^resources\|^| kernel\|^| les\|^| at::\|^| __amd" || echo "pass $i ($grp) failed: $(tail -2 $O/p$i.log | head -1)"
  rm -rf $O/p$i
done
