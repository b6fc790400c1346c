#!/usr/bin/env python
"""End-to-end run at the Adirondack-H shape (BASELINE configs[1]/[3] substitute, SURVEY.md 8(d) "End-to-end").

The Adirondack data set (1.2 GB, MC-CNN volumes) is not in the container, so a synthetic scene of the same shape is
used: piecewise-planar ground-truth disparity, textured left image, right image = left warped by the ground truth,
matching-cost volumes = truncated absolute colour differences computed with torch on the GPU (float [ndisp][H][W], the
.acrt layout).  Then the MidV3 loop of LES/main.cpp:330-420: volume ingest (fillOutOfView / convertVolumeL2R), layers
1 % / 3 % / 9 % of the width, pmIterations PatchMatch iterations + `iterations` graph-cut iterations, optional two-view
post-processing, Evaluator rows (bad-1.0).  Prints one JSON object with the wall-clock split.

  python tools/e2e_bench.py [--width 1436 --height 992 --ndisp 256 --iterations 5 --pm-iterations 2 --dual 0]
"""
import argparse
import json
import os

os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # before any OpenMP runtime starts: sleeping workers between lock-steps beat spinning ones
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


from localexpstereo_amd.synth import make_scene, make_scene_three_surfaces, ad_volume      # noqa: E402


def scene_inputs(scene, H, W, D, dev):
    """(imL, imR, gt, volL as a host array): "objects" = synth.make_scene + absolute-difference volume (nine small objects), "three_surfaces" = the
    C++ host demo's scene (DemoScene.h; large slanted surfaces: the hard one for the cuts)."""
    key = (scene, H, W, D)
    if _scene_cache.get("key") == key:                 # (one entry: bench.py runs one view and two views on the same scene back to back)
        return _scene_cache["val"]
    if scene == "three_surfaces":
        val = make_scene_three_surfaces(H, W, D)
    elif scene == "objects":
        imL, imR, gt = make_scene(H, W, D)
        val = (imL, imR, gt, ad_volume(imL, imR, D, dev).cpu().numpy())        # host arrays = what the .acrt reader hands over
    else:
        raise ValueError(f"unknown scene {scene!r}")
    _scene_cache.clear()
    _scene_cache.update(key=key, val=val)
    return val


_scene_cache = {}


def run(width=1436, height=992, ndisp=256, iterations=5, pm_iterations=2, dual=0, smooth_weight=0.5, host_threads=0, quiet=False, scene="objects", device_cuts=None):
    """One end-to-end run; returns the record `main` prints."""
    import torch  # noqa: F401
    from localexpstereo_amd import stereo
    dev = "cuda"
    H, W, D = height, width, ndisp
    t0 = time.perf_counter()
    imL, imR, gt, volL = scene_inputs(scene, H, W, D, dev)
    t_scene = time.perf_counter() - t0
    data = dict(imL=imL, imR=imR, dispGT=gt, nonocc=np.ones((H, W), bool), ndisp=D, gt_prec=-1.0)
    def cpu_stat():
        # cgroup CPU accounting: a process that keeps more threads busy than its quota grants is stopped until the next period
        try:
            return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("usage_usec", "nr_throttled", "throttled_usec")}
        except Exception:
            return {}
    c0 = cpu_stat()
    t1 = time.perf_counter()
    st, lab, raw = stereo.MidV3(data, volL, None, iterations=iterations, pmIterations=pm_iterations, doDual=bool(dual),
                                smooth_weight=smooth_weight, mc_threshold=0.5, error_threshold=1.0, device=dev, host_threads=host_threads, device_cuts=device_cuts)
    t_total = time.perf_counter() - t1
    c1 = cpu_stat()
    cpu = {k: c1[k] - c0[k] for k in c0 if k in c1}
    if "usage_usec" in cpu:
        cpu = {"cpu_seconds": round(cpu["usage_usec"] * 1e-6, 2), "mean_cpus_busy": round(cpu["usage_usec"] * 1e-6 / t_total, 2),
               "periods_throttled": cpu.get("nr_throttled"), "seconds_throttled": round(cpu.get("throttled_usec", 0) * 1e-6, 3)}
    rows = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in st.log]
    return dict(scene=scene, shape=[W, H, D], iterations=iterations, pm_iterations=pm_iterations, dual=bool(dual), host_cores=os.cpu_count(),
                seconds_total_including_ingest=round(t_total, 3), seconds_optimiser=round(st.seconds, 3), seconds_reference_clock=round(getattr(st, "seconds_reference_clock", st.seconds), 3), seconds_evaluation=round(st.eval_seconds, 3), scene_seconds=round(t_scene, 2),
                gc_seconds={k: round(v, 3) for k, v in st.gc_seconds.items()}, tiled_locksteps=getattr(st, "tiled_lockstep_stats", {}), cgroup_cpu=cpu, host_threads=host_threads, log=rows)


def run_sharded(rank, world, device, width=1436, height=992, ndisp=256, iterations=5, pm_iterations=2, smooth_weight=0.5, scene="objects"):
    """BASELINE configs[3]: the two-view run with the views split over two rank groups and the cells of every disjoint set sharded
    inside a group (stereo.FastGCStereo.run; one all-gather of the updated tiles per set, LES/FastGCStereo.h:22-72, 172-185, 199-203).
    Every rank calls it (torch.distributed is initialised by the caller); returns this rank's record."""
    import torch
    from localexpstereo_amd import stereo
    H, W, D = height, width, ndisp
    imL, imR, gt, volL = scene_inputs(scene, H, W, D, device)
    data = dict(imL=imL, imR=imR, dispGT=gt, nonocc=np.ones((H, W), bool), ndisp=D, gt_prec=-1.0)
    torch.cuda.synchronize(torch.device(device))
    t1 = time.perf_counter()
    st, lab, raw = stereo.MidV3(data, volL, None, iterations=iterations, pmIterations=pm_iterations, doDual=True, smooth_weight=smooth_weight,
                                mc_threshold=0.5, error_threshold=1.0, device=device, rank=rank, world=world)
    t_total = time.perf_counter() - t1
    bad = [r.get("all") for r in st.log if r.get("all") is not None]
    return dict(seconds_total_including_ingest=round(t_total, 3), seconds_optimiser=round(st.seconds, 3), bytes_exchanged=int(st.bytes_exchanged),
                all_gathers=int(st.all_gathers), exchange_seconds=round(float(getattr(st, "exchange_seconds", 0.0)), 4), host_cut_seconds=round(float(st.gc_seconds.get("host_cuts", st.gc_seconds.get("cuts", 0.0))), 3),
                gc_seconds={k: round(v, 3) for k, v in st.gc_seconds.items()}, bad_all_last=(round(bad[-1], 3) if bad else None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1436)
    ap.add_argument("--height", type=int, default=992)
    ap.add_argument("--ndisp", type=int, default=256)
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--pm-iterations", type=int, default=2)
    ap.add_argument("--dual", type=int, default=0)
    ap.add_argument("--smooth-weight", type=float, default=0.5)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--scene", default="objects", choices=["objects", "three_surfaces"])
    ap.add_argument("--device-cuts", default=None, choices=[None, "none", "fine", "all"], help="which layers are cut on the GPU (default: all that the library supports)")
    args = ap.parse_args()
    print(json.dumps(run(args.width, args.height, args.ndisp, args.iterations, args.pm_iterations, args.dual, args.smooth_weight, args.host_threads, scene=args.scene, device_cuts=args.device_cuts)))


if __name__ == "__main__":
    main()
