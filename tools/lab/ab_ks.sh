# K x S sweep of the tiled max-flow (inner iterations per launch x sweeps between exact relabellings, all rounds) on whole one-view runs
O=${1:-gpurun_out/ab_ks}; mkdir -p $O
for cfg in "8 12" "4 12" "4 16" "4 24" "2 24" "2 32" "3 16" "3 24" "6 12" "6 16"; do
set -- $cfg; K=$1; S=$2
for sc in objects three_surfaces; do
  LES_HIP_MAXFLOW_TILED_K=$K LES_HIP_MAXFLOW_TILED_S=$S timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_K${K}_S${S}.json 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k == "tiled_launches"}, {k: (v["ms_p50"], v["launches_p50"]) for k, v in d["tiled_locksteps"].items()})
PY
