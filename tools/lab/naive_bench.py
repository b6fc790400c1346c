#!/usr/bin/env python
"""Throughput of the image-based matching cost (NaiveStereoEnergy, the MiddV2 mode's energy) on the device: the march kernel fed by
the raw-cost pre-pass (les_naive_raw_kernel + role A's one-tap path) against the fp64 strip kernel with SRC = 1 (LES_HIP_KERNEL=strip),
same batches, same planes.  Two shapes: whole-image slabs (N planes x W x H) and the LayerManager cells of layer 0 (one launch per
disjoint set).  Each kernel runs in its own process (the choice is made when the context is created).

  python tools/naive_bench.py [--W 1500 --H 1000 --planes 64 --steps 10]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    import torch
    from localexpstereo_amd import api, pm, synth
    dev = torch.device("cuda:0")
    H, W, N = args.H, args.W, args.planes
    imL, imR = synth.make_guide(H, W, 11), synth.make_guide(H, W, 12)
    e = api.HipCostVolumeEnergy.naive(imL, imR, windR=20, eps=1e-4, max_disp=255.0)
    rng = np.random.default_rng(5)
    out = {"kernel": os.environ.get("LES_HIP_KERNEL", "march")}

    def timed(fn, steps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    # whole-image slabs
    frs = np.tile(np.array([[0, 0, W, H]], np.int32), (N, 1))
    planes = np.zeros((N, 4), np.float32)
    planes[:, 0] = rng.uniform(-0.05, 0.05, N)
    planes[:, 1] = rng.uniform(-0.05, 0.05, N)
    planes[:, 2] = rng.uniform(0, 200, N)
    bt = api.Batch(e, frs, frs, out_slabs=True)
    buf = torch.empty((N, H, W), device=dev, dtype=torch.float32)
    pl = torch.from_numpy(planes).to(dev)
    ms = timed(lambda: bt.run(pl.data_ptr(), buf.data_ptr(), mode=0, check=True, planes_on_device=True), args.steps)
    out["slabs"] = {"ms": round(ms, 3), "Mevals_per_s": round(N * H * W / ms / 1e3, 1), "kernel_kind": bt.kernel_kind(0), "sum": float(buf.double().sum())}
    bt.destroy()
    del buf
    # layer-0 cells, one launch per disjoint set
    units, shared, filt, sets = pm.layer_geometry(W, H, 20, max(1, int(W * 0.01)))
    cmap = torch.empty((H, W), device=dev, dtype=torch.float32)
    batches, pls = [], []
    for s in sets:
        batches.append(api.Batch(e, filt[s], shared[s]))
        p = np.zeros((len(s), 4), np.float32)
        p[:, 0] = rng.uniform(-0.05, 0.05, len(s))
        p[:, 2] = rng.uniform(0, 200, len(s))
        pls.append(torch.from_numpy(p).to(dev))
    f4 = filt.view(np.int32).reshape(-1, 4)
    evals = int(sum((f4[s][:, 2].astype(np.int64) * f4[s][:, 3]).sum() for s in sets))

    def cells():
        for b, p in zip(batches, pls):
            b.run(p.data_ptr(), cmap.data_ptr(), mode=0, check=True, planes_on_device=True)

    ms = timed(cells, args.steps)
    out["cells_layer0"] = {"ms": round(ms, 3), "launch_sets": len(sets), "filter_domain_Mevals_per_s": round(evals / ms / 1e3, 1), "kernel_kind": batches[0].kernel_kind(0)}
    for b in batches:
        b.destroy()
    e.close()
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--W", type=int, default=1500)
    ap.add_argument("--H", type=int, default=1000)
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    res = []
    for kern in ("march", "strip"):
        env = dict(os.environ)
        if kern == "strip":
            env["LES_HIP_KERNEL"] = "strip"
        else:
            env.pop("LES_HIP_KERNEL", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--W", str(args.W), "--H", str(args.H), "--planes", str(args.planes),
                            "--steps", str(args.steps)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(p.stdout[-2000:], p.stderr[-2000:])
            raise SystemExit(1)
        res.append(json.loads(line[0][7:]))
    a, b = res[0]["slabs"]["sum"], res[1]["slabs"]["sum"]
    print(json.dumps({"shape": [args.W, args.H, args.planes], "march": res[0], "strip": res[1], "slab_sum_rel_diff": abs(a - b) / max(abs(b), 1e-30)}))


if __name__ == "__main__":
    main()
