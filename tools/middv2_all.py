#!/usr/bin/env python
"""The reference's MidV2 mode (LES/main.cpp:270-328) with its DEFAULTS -- iterations 5, pmIterations 2, one view, smooth weight 1,
filter radius 20, ndisp from info.txt, layers 5 / 15 / 25, Evaluator bad-0.5 on disparities quantised to the ground-truth precision
-- on the four Middlebury-2003 pairs the reference bundles under data/MiddV2 (copied as data fixtures to tests/golden/).  The oracle
cannot be pinned to a run of the reference (it is unbuildable here), so these real-data trajectories are the anchor: per-iteration
energy and error rates are written to gpurun_out/middv2_all.json (-> profiles/round3_middv2.json) and asserted with tight per-set
bounds in tests/test_gpu_parity.py::test_gpu_middv2_all_sets_reference_defaults.

  python tools/middv2_all.py [--sets cones,teddy,venus,tsukuba] [--dual 0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def run_set(name, dual=False, iterations=5, pm_iterations=2):
    from localexpstereo_amd import io as lio
    from localexpstereo_amd import stereo
    data = lio.load_data(os.path.join(ROOT, "tests", "golden", name))          # ndisp and ground-truth scale from info.txt
    t0 = time.perf_counter()
    st, lab, raw = stereo.MidV2(data, iterations=iterations, pmIterations=pm_iterations, doDual=dual)
    wall = time.perf_counter() - t0
    rows = [{k: (round(float(v), 4) if isinstance(v, float) else v) for k, v in r.items()} for r in st.log]
    return dict(set=name, shape=[int(data["imL"].shape[1]), int(data["imL"].shape[0]), int(data["ndisp"])], dual=bool(dual), iterations=iterations,
                pm_iterations=pm_iterations, seconds=round(wall, 3), gc_seconds={k: round(v, 3) for k, v in st.gc_seconds.items()}, log=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="cones,teddy,venus,tsukuba")
    ap.add_argument("--dual", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "middv2_all.json"))
    args = ap.parse_args()
    res = []
    for name in args.sets.split(","):
        r = run_set(name, bool(args.dual))
        res.append(r)
        last = r["log"][-1]
        print(f"{name:8s} {r['shape']}  {r['seconds']:.2f} s   final E={last['energy']:.1f}  all={last['all']:.2f}%  nonocc={last['nonocc']:.2f}%", flush=True)
        for row in r["log"]:
            print("   ", row["index"], row["time"], "E", row["energy"], "data", row["data"], "smooth", row["smooth"], "all", row["all"], "nonocc", row["nonocc"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
