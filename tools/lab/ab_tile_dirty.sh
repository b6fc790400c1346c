# Tiled max-flow, round 6 lab patch (not adopted, DESIGN 3.4: no effect): DISCHARGE writes back only the nodes whose residuals or excess changed (product) against the whole tile (libles_storeall.so: tools/build_variant.sh storeall -DLES_MT_STORE_ALL)
O=${1:-gpurun_out/ab_dirty}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in storeall product; do
  if [ $v = product ]; then unset LES_HIP_LIB; else export LES_HIP_LIB=localexpstereo_amd/csrc/libles_$v.so; [ -f $LES_HIP_LIB ] || continue; fi
  for sc in objects three_surfaces; do
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_$v.json 2>$O/err.log
    python tools/prof_summary.py $O/prof --md 2>/dev/null | grep "les_maxflow_tiled_kernel" | head -1 | cut -c1-50,60-120
    rm -rf $O/prof
    timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_$v.json 2>$O/err.log
  done
done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    d = json.loads(open(f).read()); g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], "energy", round(d["log"][-1]["energy"], 3), {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_launches",)}, {k: v["ms_p50"] for k, v in d["tiled_locksteps"].items()})
PY
