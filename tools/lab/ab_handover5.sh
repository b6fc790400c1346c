# Hand-over of stragglers, final tree of round 6: launches before the first hand-over (product 28) x open cells at most (product 8), two-view runs of both scenes
O=${1:-gpurun_out/ab_ho5}; mkdir -p $O
for cfg in "28 8" "20 8" "36 8" "44 8" "28 12" "28 16" "28 5" "36 5" "44 12"; do
set -- $cfg
for sc in objects three_surfaces; do
  LES_HIP_MAXFLOW_HANDOVER_AFTER=$1 LES_HIP_MAXFLOW_HANDOVER_CELLS=$2 timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > "$O/e2e_${sc}_dual_a$1_c$2.json" 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    d = json.loads(open(f).read()); g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_handed_cells", "tiled_handed_host_seconds", "tiled_launches")}, {k: v["ms_p50"] for k, v in d["tiled_locksteps"].items() if k.startswith("view1")})
PY
