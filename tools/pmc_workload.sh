#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, calibrated with the known-size dword copy of the same run) and SQ instruction
# counters of the march kernel on ANOTHER workload of the bench (h2: 256 slanted planes; h3: the optimiser's cell batches), per launch:
#   bash tools/pmc_workload.sh h2 [tag]      -> gpurun_out/prof/<tag>_<workload>_pmc.md     (run on the GPU box through gpurun)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-h2}; TAG=${2:-round6}
O=gpurun_out/prof; mkdir -p $O
B="python bench.py --workload $WL --steps 3 --warmup 1 --cpu-planes 0 --sub-steps 0 --e2e 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/w_$c -- $B > $O/w_$c.log 2>&1
  python tools/prof_summary.py $O/w_$c les_march_kernel --md > $O/w_$c.md
  rocprofv3 --pmc $c -d $O/wcal_$c -- python tools/calib_copy.py > $O/wcal_$c.log 2>&1
  python tools/prof_summary.py $O/wcal_$c les_calib --md > $O/wcal_$c.md
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/w_sq -- $B > $O/w_sq.log 2>&1
python tools/prof_summary.py $O/w_sq les_march_kernel --md > $O/w_sq.md
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/w_tcc -- $B > $O/w_tcc.log 2>&1
python tools/prof_summary.py $O/w_tcc les_march_kernel --md > $O/w_tcc.md
$B > $O/w_bench.json 2>/dev/null
{
  echo "# ${TAG}: march kernel on workload ${WL} -- counters per launch (separate rocprofv3 --pmc passes of \`$B\`)"
  echo; echo "## FETCH_SIZE (KiB)"; cat $O/w_FETCH_SIZE.md; echo; echo "calibration copy (1.536 GB read + 1.536 GB written per launch):"; cat $O/wcal_FETCH_SIZE.md
  echo; echo "## WRITE_SIZE (KiB)"; cat $O/w_WRITE_SIZE.md; echo; echo "calibration copy:"; cat $O/wcal_WRITE_SIZE.md
  echo; echo "## SQ instruction counters"; cat $O/w_sq.md
  echo; echo "## L2 (TCC) requests"; cat $O/w_tcc.md
  echo; echo "## the bench line of the same workload"; python -c "
import json,sys; d=json.loads(open('$O/w_bench.json').readline()); print({k: d[k] for k in ('ms_per_step','value','unit')}, d['config']['workload'][:160], {k: d['roofline'][k] for k in ('achieved','frac','kernel_ms','algorithmic_bytes_per_launch')})"
} > $O/${TAG}_${WL}_pmc.md
rm -rf $O/w_FETCH_SIZE $O/w_WRITE_SIZE $O/wcal_FETCH_SIZE $O/wcal_WRITE_SIZE $O/w_sq $O/w_tcc
tail -60 $O/${TAG}_${WL}_pmc.md
