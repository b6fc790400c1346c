# Two views on one GPU: the slower view on a high-priority stream (a stereo.py patch of round 6 read LES_VIEW_PRIORITY; no effect, not adopted: DESIGN 3.4) against both at the default priority; whole runs, twice each
O=${1:-gpurun_out/ab_prio}; mkdir -p $O
for rep in 1 2; do for sc in objects three_surfaces; do for pr in 0 1; do
  LES_VIEW_PRIORITY=$pr timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_prio${pr}_$rep.json 2>$O/err.log
done; done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    d = json.loads(open(f).read()); g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], "energy", [round(l["energy"], 3) for l in d["log"][-2:]], {k: v["ms_sum"] for k, v in d["tiled_locksteps"].items()})
PY
