import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth
H, W, D = 992, 1436, 256
imL, imR, gt = synth.make_scene(H, W, D)
vol = torch.rand((2, H, W), device="cuda")
e = api.HipCostVolumeEnergy(imL, imR, vol.data_ptr(), vol.data_ptr(), max_disp=D - 1.0, volumes_on_device=True, shape=(2, H, W))
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
rng = np.random.default_rng(0)
def labels(d):
    lab = np.zeros((H, W, 4), np.float32); lab[..., 2] = d
    bad = rng.random((H, W)) < 0.05
    lab[bad, 2] += rng.uniform(5, 30, int(bad.sum())).astype(np.float32)
    return torch.from_numpy(lab).cuda()
# right-view disparity: sample gt at x + d (approx) -> use gt shifted crudely
gtr = np.zeros_like(gt)
xr = np.clip(np.rint(xs - gt).astype(int), 0, W - 1)
gtr[ys.astype(int), xr] = gt
gtr[gtr == 0] = gt[gtr == 0]
LL, LR = labels(gt), labels(gtr)
for it in range(3):
    a, b = LL.clone(), LR.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.post_process(a.data_ptr(), b.data_ptr(), 1.5, 10.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    changed = float((a != LL).any(dim=2).float().mean())
    print("post_process %dx%d: %.1f ms, %.1f %% of left labels changed" % (W, H, 1e3 * (t1 - t0), 100 * changed))
