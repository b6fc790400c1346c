# Finest layer's LDS max-flow: global relabelling every G iterations (compile-time LES_MF_G, product 8): variants built with tools/build_variant.sh gN -DLES_MF_G=N
O=${1:-gpurun_out/ab_mf_g}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in product g3 g4 g16; do
  if [ $v = product ]; then unset LES_HIP_LIB; else export LES_HIP_LIB=localexpstereo_amd/csrc/libles_$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$v -- python tools/e2e_bench.py > $O/e2e_$v.json 2>$O/err_$v.log
  python tools/prof_summary.py $O/prof_$v --md 2>/dev/null | grep "les_maxflow_kernel" | head -1 | cut -c1-40,120-180
  rm -rf $O/prof_$v
  python -c "
import json,sys
d=json.loads(open('$O/e2e_$v.json').read()); print('$v', 'optimiser', d['seconds_optimiser'], 'bad1.0', d['log'][-1]['all'], 'energy', d['log'][-1]['energy'])"
done
