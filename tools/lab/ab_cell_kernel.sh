# Finest layer: les_maxflow_cell_kernel (two-barrier iteration, residuals in registers) against les_maxflow_kernel, and its iterations per round
O=${1:-gpurun_out/ab_cell}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "0 16" "1 16" "1 8" "1 32" "1 64"; do
set -- $cfg
  LES_HIP_MAXFLOW_CELL_KERNEL=$1 LES_HIP_MAXFLOW_ROUND_ITERS=$2 timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/e2e_bench.py > $O/e2e_$1_$2.json 2>$O/err.log
  python tools/prof_summary.py $O/prof --md 2>/dev/null | grep "les_maxflow_kernel\|les_maxflow_cell" | head -2 | cut -c1-60,100-200
  rm -rf $O/prof
  python -c "
import json
d=json.loads(open('$O/e2e_$1_$2.json').read()); print('cell kernel $1 round iters $2', 'optimiser', d['seconds_optimiser'], 'bad1.0', d['log'][-1]['all'], 'energy', d['log'][-1]['energy'], d['gc_seconds'].get('cells_recut_on_host'), d['gc_seconds'].get('sets_rolled_back'))"
done
