# A/B of the tiled max-flow's FIRST-round schedule (K inner iterations x S sweeps before the first exact relabelling; product 8 x 12) on whole runs
O=${1:-gpurun_out/r6k}; mkdir -p $O
for cfg in "8 12" "8 2" "8 3" "8 4" "8 6" "4 6" "4 12" "16 2"; do
set -- $cfg; K=$1; S=$2
for sc in objects three_surfaces; do
  LES_HIP_MAXFLOW_TILED_K=$K LES_HIP_MAXFLOW_TILED_S=$S LES_HIP_MAXFLOW_TILED_K2=8 LES_HIP_MAXFLOW_TILED_S2=12 timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_K${K}_S${S}.json 2>$O/err.log
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
