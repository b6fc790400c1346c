"""Where does the slanted-plane workload (H2) spend its extra time?  Times whole-image batches of 256 planes on the 1500x1000x256
volume for plane families that separate the effects: integer fronto (one tap, KIND 0), fractional fronto (two taps, KIND 1),
the general path on (almost) fronto planes (KIND 2 arithmetic, H1's memory pattern), and growing slopes."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth
H, W, D = 1000, 1500, 256
dev = torch.device("cuda", 0)
guide = synth.make_guide(H, W, 1234)
vol = torch.rand((D, H, W), device=dev, dtype=torch.float32)
e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1, volumes_on_device=True, shape=(D, H, W))
out = torch.empty((D, H, W), device=dev, dtype=torch.float32)
full = [(0, 0, W, H)] * D
b = api.Batch(e, full, full, out_slabs=True)
def run(name, planes):
    p = torch.from_numpy(np.ascontiguousarray(planes, np.float32)).to(dev)
    for _ in range(3):
        b.run(p.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        b.run(p.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t) / n * 1e3:.3f} ms")
k = np.arange(D, dtype=np.float32)
z = np.zeros(D, np.float32)
run("fronto, integer disparity (KIND 0)", np.stack([z, z, k, z], 1))
run("fronto, fractional disparity (KIND 1, two taps)", np.stack([z, z, np.minimum(k + 0.37, D - 1.01), z], 1))
run("general path, slope 1e-9 (KIND 2 arithmetic, one slice)", np.stack([z + 1e-9, z, np.minimum(k + 0.37, D - 1.01), z], 1))
for a in (0.001, 0.005, 0.02, 0.05):
    rng = np.random.default_rng(3)
    A = rng.uniform(-a, a, D).astype(np.float32); B = rng.uniform(-a, a, D).astype(np.float32)
    C = (rng.uniform(0.3, 0.7, D) * (D - 1) - A * W / 2 - B * H / 2).astype(np.float32)
    run(f"slanted, |a|,|b| <= {a}", np.stack([A, B, C, z], 1))
run("synth.slanted_planes (bench H2)", synth.slanted_planes(D, H, W, D - 1, seed=7))
