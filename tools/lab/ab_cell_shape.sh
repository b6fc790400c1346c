# Finest layer's les_maxflow_cell_kernel: threads x nodes per thread (product 1024 x 2).  Variants: tools/build_variant.sh mc512 -DLES_MC_THREADS=512 -DLES_MC_NPT=4 etc.
O=${1:-gpurun_out/ab_cell_shape}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS:-product mc512 mc768 mc256}; do
  if [ $v = product ]; then unset LES_HIP_LIB; else export LES_HIP_LIB=localexpstereo_amd/csrc/libles_$v.so; [ -f $LES_HIP_LIB ] || continue; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/e2e_bench.py > $O/e2e_$v.json 2>$O/err.log
  python tools/prof_summary.py $O/prof --md 2>/dev/null | grep "les_maxflow_cell" | head -1 | cut -c1-60,100-200
  rm -rf $O/prof
  python -c "
import json
d=json.loads(open('$O/e2e_$v.json').read()); print('$v', 'optimiser', d['seconds_optimiser'], 'bad1.0', d['log'][-1]['all'], 'energy', d['log'][-1]['energy'], d['gc_seconds'].get('cells_recut_on_host'))"
done
