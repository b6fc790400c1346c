#!/bin/bash
# SQ / LDS counters of the march kernel on the headline bench (run on the GPU box through gpurun).  Separate --pmc passes, no trace domains.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc; mkdir -p $O
B="python bench.py --steps 3 --warmup 1 --cpu-planes 0 --sub-steps 0"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
           "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/p$i -- $B > $O/p$i.log 2>&1
  python tools/prof_summary.py $O/p$i les_march --md > $O/p$i.md 2>/dev/null || python tools/prof_summary.py $O/p$i les_strip --md > $O/p$i.md
  rm -rf $O/p$i
done
cat $O/p*.md | grep -v "^##\|^|---\|^| counter"
