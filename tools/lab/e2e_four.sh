# The four end-to-end numbers (two scenes, one view / two views) with the per-layer seconds of the tiled solver.  Usage (GPU box): bash tools/lab/e2e_four.sh <outdir>
O=${1:-gpurun_out/e2e4}; mkdir -p $O
for sc in objects three_surfaces; do
  timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual.json 2>$O/err.log
  timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single.json 2>$O/err.log
done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_launches", "tiled_handed_cells")}, {k: (v["ms_p50"], v["launches_p50"]) for k, v in d["tiled_locksteps"].items()}, "bad1.0", d["log"][-1]["all"])
PY
