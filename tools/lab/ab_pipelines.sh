O=gpurun_out/r6h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "per_cell_pipelines or without_round_trips" > $O/test.log 2>&1; tail -3 $O/test.log
for sc in objects three_surfaces; do for lo in 0 1; do
  if [ $lo = 1 ]; then export LES_GC_LOCKSTEP_ORDER=1; else unset LES_GC_LOCKSTEP_ORDER; fi
  timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_lock$lo.json 2>$O/err_${sc}_dual_$lo.log
  timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_lock$lo.json 2>$O/err_${sc}_single_$lo.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], "total", d["seconds_total_including_ingest"], {k: round(g[k], 2) for k in g if k.startswith("tiled_h") or k.startswith("tiled_sec") or k.startswith("sets_")}, d.get("evaluator"))
PY
