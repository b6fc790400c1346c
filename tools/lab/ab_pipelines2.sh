O=gpurun_out/r6i; mkdir -p $O
unset LES_GC_LOCKSTEP_ORDER
for cfg in "HANDOVER=0" "HANDOVER_AFTER=60" "HANDOVER_AFTER=100" "HANDOVER_AFTER=160"; do
for sc in objects three_surfaces; do
  env LES_HIP_MAXFLOW_$cfg timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_$cfg.json 2>$O/err.log
  env LES_HIP_MAXFLOW_$cfg timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_$cfg.json 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_h") or k.startswith("tiled_sec") or k.startswith("sets_as")})
PY
