"""Fixed cost of a march launch against its cost per tick: whole-image slabs of 224 planes (one round of workgroups: 224 x ceil(W/216) <= 256 for W = 216)
on images of growing height, fronto-parallel and slanted planes; a least-squares line through (ticks, microseconds) gives the time per tick and what a
launch costs before its first tick pays (cold start of the loads + the three pipeline ticks are in `ticks`).   python tools/tick_fit.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth
dev = torch.device("cuda", 0)
W, D, NP = 216, 64, 224
for fam in ("fronto", "slanted |a|,|b| <= 0.05", "slanted |a| = 0.3"):
    pts = []
    for H in (64, 128, 256, 512, 1024):
        guide = synth.make_guide(H, W, 7)
        vol = torch.rand((D, H, W), device=dev, dtype=torch.float32)
        e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1, volumes_on_device=True, shape=(D, H, W))
        full = [(0, 0, W, H)] * NP
        b = api.Batch(e, full, full, out_slabs=True)
        rng = np.random.default_rng(1)
        z = np.zeros(NP, np.float32)
        if fam == "fronto":
            pl = np.stack([z, z, rng.integers(0, D, NP).astype(np.float32), z], 1)
        else:
            s = 0.05 if "0.05" in fam else 0.3
            A = (rng.uniform(-s, s, NP) if s == 0.05 else rng.choice([-s, s], NP)).astype(np.float32); B = rng.uniform(-0.05, 0.05, NP).astype(np.float32)
            pl = np.stack([A, B, (D / 2 - A * W / 2 - B * H / 2).astype(np.float32), z], 1)
        p = torch.from_numpy(np.ascontiguousarray(pl, np.float32)).to(dev)
        out = torch.empty((NP, H, W), device=dev, dtype=torch.float32)
        for _ in range(3):
            b.run(p.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
        torch.cuda.synchronize()
        n = 30
        t = time.perf_counter()
        for _ in range(n):
            b.run(p.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t) / n * 1e6
        ticks = (H + 40 + 6) // 7 + 3
        pts.append((ticks, us, int(b.num_jobs) if hasattr(b, "num_jobs") else -1))
        b.destroy(); e.close()
    x = np.array([q[0] for q in pts], float); y = np.array([q[1] for q in pts], float)
    k, c = np.polyfit(x, y, 1)
    print(f"{fam:28s} " + "  ".join(f"{int(a)} ticks {b_:.1f} us" for a, b_, _ in pts) + f"   -> {k:.3f} us per tick, {c:.1f} us fixed per launch")
