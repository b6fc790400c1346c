# Large open sets (the coarsest layer's 150 000-node cells) after the launches became cheaper: is handing them over still worth it?
O=${1:-gpurun_out/ab_ho3}; mkdir -p $O
for cfg in "LATE_AFTER=220 ALL_AFTER=300" "LATE_AFTER=100000 ALL_AFTER=300" "LATE_AFTER=100000 ALL_AFTER=450" "LATE_AFTER=100000 ALL_AFTER=700" "LATE_AFTER=100000 ALL_AFTER=100000"; do
set -- $cfg
for sc in objects three_surfaces; do
  env LES_HIP_MAXFLOW_HANDOVER_$1 LES_HIP_MAXFLOW_HANDOVER_$2 timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > "$O/e2e_${sc}_dual_$1_$2.json" 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_handed_cells", "tiled_handed_host_seconds")}, {k: v["ms_max"] for k, v in d["tiled_locksteps"].items()})
PY
