#!/bin/bash
# Round 6: same-box A/B of the tiled max-flow's hand-over on whole runs (two scenes, one and two views), then the kernel-trace stats of a one-view run.
# Usage (GPU box): bash tools/r6_e2e_ab.sh <outdir>
O=${1:-gpurun_out/r6_ab}; mkdir -p $O
for sc in objects three_surfaces; do for ho in 1 0; do
  LES_HIP_MAXFLOW_HANDOVER=$ho timeout 150 python tools/e2e_bench.py --dual 1 --scene $sc > $O/e2e_${sc}_dual_ho$ho.json 2>/dev/null
  LES_HIP_MAXFLOW_HANDOVER=$ho timeout 100 python tools/e2e_bench.py --scene $sc > $O/e2e_${sc}_single_ho$ho.json 2>/dev/null
done; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_e2e -- python tools/e2e_bench.py > $O/prof_e2e.log 2>&1
python tools/prof_summary.py $O/prof_e2e --md > $O/e2e_kernel_stats.md
rm -rf $O/prof_e2e
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/e2e_*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as ex:
        print(f, "unreadable", ex); continue
    g = d["gc_seconds"]
    print(f.split("/")[-1], "optimiser", d["seconds_optimiser"], "total", d["seconds_total_including_ingest"], {k: round(g[k], 2) for k in g if k.startswith("tiled_h") or k.startswith("tiled_sec")})
    for k, v in d["tiled_locksteps"].items():
        print("    ", k, {a: v[a] for a in ("ms_p50", "ms_p90", "ms_max", "ms_sum", "launches_p50")})
PY
head -24 $O/e2e_kernel_stats.md
