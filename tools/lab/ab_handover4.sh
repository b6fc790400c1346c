# early hand-over only of small cells (LES_HIP_MAXFLOW_HANDOVER_CELL_NODES; 10^9 = the rule of before)
O=${1:-gpurun_out/ab_ho4}; mkdir -p $O
for cn in 40000 1000000000; do
for sc in objects three_surfaces; do
  LES_HIP_MAXFLOW_HANDOVER_CELL_NODES=$cn timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > "$O/e2e_${sc}_dual_$cn.json" 2>$O/err.log
  LES_HIP_MAXFLOW_HANDOVER_CELL_NODES=$cn timeout 150 python tools/e2e_bench.py --scene $sc > "$O/e2e_${sc}_single_$cn.json" 2>$O/err.log
done; done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    d = json.loads(open(f).read()); g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], {k: round(g[k], 2) for k in g if k.startswith("tiled_sec") or k in ("tiled_handed_cells", "tiled_handed_host_seconds")})
PY
