# Large open sets: hand over earlier AND cut the residual graph in row bands (LES_GC_RESIDUAL_BAND_NODES nodes per band; product: one band, after 220 launches)
O=${1:-gpurun_out/ab_lb}; mkdir -p $O
for cfg in "220 0" "60 20000" "92 20000" "124 20000" "60 40000" "92 40000" "220 20000"; do
set -- $cfg; LA=$1; BN=$2
for sc in objects three_surfaces; do
  if [ $BN = 0 ]; then unset LES_GC_RESIDUAL_BAND_NODES; else export LES_GC_RESIDUAL_BAND_NODES=$BN; fi
  LES_HIP_MAXFLOW_HANDOVER_LATE_AFTER=$LA timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_la${LA}_bn${BN}.json 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    d = json.loads(open(f).read()); g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_handed_cells", "tiled_handed_host_seconds")})
PY
