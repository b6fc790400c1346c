# parity of the march kernel at the per-rank shape of bench.py --gpus N (3000 x 2000), two whole-image planes against the oracle
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import oracle as om
H, W, D = 2000, 3000, 6
guide = synth.make_guide(H, W, 1234)
vol = synth.make_volume(D, H, W, 42)
e = api.HipCostVolumeEnergy(guide, None, vol, None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1)
o = om.Oracle(guide, guide, vol, vol, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1)
planes = np.array([[0, 0, 2.0, 0], [0.0007, -0.0005, 2.3, 0]], np.float32)
full = [(0, 0, W, H)] * 2
b = api.Batch(e, full, full, out_slabs=True)
print("kernel kind", b.kernel_kind(0), "jobs", b.num_jobs)
out = torch.empty((2, H, W), device="cuda", dtype=torch.float32)
pl = torch.from_numpy(planes).cuda()
b.run(pl.data_ptr(), out.data_ptr(), mode=0, check=True, planes_on_device=True)
torch.cuda.synchronize()
got = out.cpu().numpy()
t = time.time()
for k in range(2):
    ref = o.unary_batch([(0, 0, W, H)], [(0, 0, W, H)], planes[k][None], check=True)
    err = np.abs(got[k].astype(np.float64) - ref.astype(np.float64))
    print("plane", k, "max abs err", err.max(), "oracle s", round(time.time() - t, 1))
    assert err.max() < 2.5e-5
print("OK")
