"""Taps of slanted planes from the tiled copy of the volume ([H][W/8][D][8], role A's KIND 5) against taps from [D][H][W] (KIND 4): ms per
pass of 256 whole-image planes on the 1500x1000x256 volume, by x-slope |a| -- where the tiled copy starts to pay (LES_TILED_MIN_SLOPE), and
that both give the same costs bit for bit.   python tools/tiled_probe.py      (GPU box)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth
H, W, D = 1000, 1500, 256
dev = torch.device("cuda", 0)
guide = synth.make_guide(H, W, 1234)
vol = torch.rand((D, H, W), device=dev, dtype=torch.float32)
out = torch.empty((D, H, W), device=dev, dtype=torch.float32)
full = [(0, 0, W, H)] * D
fams = []
z = np.zeros(D, np.float32)
for s in (0.02, 0.05, 0.08, 0.1, 0.125, 0.15, 0.25, 0.5):
    rng = np.random.default_rng(3)
    A = (rng.choice([-1.0, 1.0], D) * s).astype(np.float32); B = rng.uniform(-0.05, 0.05, D).astype(np.float32)
    C = (rng.uniform(0.3, 0.7, D) * (D - 1) - A * W / 2 - B * H / 2).astype(np.float32)
    fams.append((f"|a| = {s}", np.stack([A, B, C, z], 1)))
fams.append(("synth.slanted_planes (bench H2)", synth.slanted_planes(D, H, W, D - 1, seed=7)))
res = {}
for tiled in ("0", "1"):
    os.environ["LES_HIP_TILED"] = tiled
    os.environ["LES_TILED_PROBE"] = "1"
    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1, volumes_on_device=True, shape=(D, H, W))
    b = api.Batch(e, full, full, out_slabs=True)
    for name, planes in fams:
        p = torch.from_numpy(np.ascontiguousarray(planes, np.float32)).to(dev)
        for _ in range(2):
            b.run(p.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 10
        for _ in range(n):
            b.run(p.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / n * 1e3
        chk = out[::37].clone()
        res.setdefault(name, []).append((ms, chk))
    b.destroy(); e.close()
for name, r in res.items():
    same = bool(torch.equal(r[0][1], r[1][1]))
    print(f"{name:36s} [D][H][W] {r[0][0]:.3f} ms | tiled copy allowed {r[1][0]:.3f} ms | bit-equal {same}")
