# Later rounds of the tiled max-flow (few active nodes: K2 inner iterations x S2 sweeps; product 8 x 12, first round fixed at 8 x 12) on whole runs
O=${1:-gpurun_out/ab_k2s2}; mkdir -p $O
for cfg in "8 12" "16 6" "12 8" "16 12" "24 4" "6 16" "4 24"; do
set -- $cfg; K=$1; S=$2
  LES_HIP_MAXFLOW_TILED_K2=$K LES_HIP_MAXFLOW_TILED_S2=$S timeout 150 python tools/e2e_bench.py --dual 1 --scene objects > $O/e2e_objects_dual_K${K}_S${S}.json 2>$O/err.log
  LES_HIP_MAXFLOW_TILED_K2=$K LES_HIP_MAXFLOW_TILED_S2=$S timeout 100 python tools/e2e_bench.py --scene three_surfaces > $O/e2e_three_surfaces_single_K${K}_S${S}.json 2>$O/err.log
done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    d = json.loads(open(f).read()); g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_launches", "tiled_handed_host_seconds")})
PY
