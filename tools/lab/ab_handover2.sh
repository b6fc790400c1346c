# Hand-over thresholds again after the launches became cheaper (two-view runs: the right view is where cells are handed over)
O=${1:-gpurun_out/ab_ho2}; mkdir -p $O
for cfg in "X=0" "HANDOVER=0" "HANDOVER_AFTER=44" "HANDOVER_AFTER=60" "HANDOVER_LATE_AFTER=140" "HANDOVER_LATE_AFTER=300" "HANDOVER_CELLS=4" "HANDOVER_CELLS=12"; do
for sc in objects three_surfaces; do
  env LES_HIP_MAXFLOW_$cfg timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_$cfg.json 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_handed_cells", "tiled_handed_host_seconds")})
PY
