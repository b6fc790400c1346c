import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth, pm
H, W, D = 992, 1436, 64
imL = synth.make_guide(H, W, 1)
vol = torch.rand((D, H, W), device="cuda")
e = api.HipCostVolumeEnergy(imL, None, vol.data_ptr(), None, max_disp=D - 1.0, volumes_on_device=True, shape=(D, H, W))
r = pm.PMRunner(e, (14,), [[(api.PROPOSE_EXPANSION, 1)]], device="cuda")
lab = np.zeros((H, W, 4), np.float32); lab[..., 2] = np.random.default_rng(0).uniform(5, 50, (H, W))
torch.cuda.synchronize(); t0 = time.perf_counter()
r.init_from_labels(lab)
torch.cuda.synchronize(); print("warm start %dx%d: %.2f s" % (W, H, time.perf_counter() - t0), float(r.cur.mean()))
