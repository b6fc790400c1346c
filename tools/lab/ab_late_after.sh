# A/B of the tiled max-flow's hand-over threshold for LARGE open sets (LES_HIP_MAXFLOW_HANDOVER_LATE_AFTER; product: 220 launches) on whole runs
O=${1:-gpurun_out/r6j}; mkdir -p $O
for la in 220 60 90 120 160; do
for sc in objects three_surfaces; do
  LES_HIP_MAXFLOW_HANDOVER_LATE_AFTER=$la timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_la$la.json 2>$O/err.log
  LES_HIP_MAXFLOW_HANDOVER_LATE_AFTER=$la timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_la$la.json 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_h") or k.startswith("tiled_sec")})
PY
