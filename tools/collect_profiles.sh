#!/bin/bash
# Collects the round's rocprofv3 evidence for the headline bench (run on the GPU box through gpurun):
#   kernel-trace stats of `python bench.py`, FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no trace domains mixed
#   in), and the same two counters on a known-byte-count dword copy for calibration.  Summaries land in gpurun_out/prof/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
B="python bench.py --steps 5 --warmup 1 --cpu-planes 0"
rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.log 2>&1
python tools/prof_summary.py $O/stats --md > $O/stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
  python tools/prof_summary.py $O/pmc_$c les_strip --md > $O/pmc_$c.md
  rocprofv3 --pmc $c -d $O/cal_$c -- python tools/calib_copy.py > $O/cal_$c.log 2>&1
  python tools/prof_summary.py $O/cal_$c les_calib --md > $O/cal_$c.md
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
python tools/prof_summary.py $O/pmc_sq les_strip --md > $O/pmc_sq.md
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE $O/pmc_sq
cat $O/stats.md | head -8; grep -h "FETCH_SIZE\|WRITE_SIZE" $O/pmc_*.md $O/cal_*.md; grep -h "SQ_" $O/pmc_sq.md
