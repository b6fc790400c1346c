"""Per-layer time of the H3 workload (LayerManager cell batches at 1500x1000x256): launches, workgroups, ms, filter-domain G evals/s."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from localexpstereo_amd import api, synth, pm
H, W, D = 1000, 1500, 256
dev = torch.device("cuda", 0)
guide = synth.make_guide(H, W, 1234)
vol = torch.rand((D, H, W), device=dev, dtype=torch.float32)
e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1, volumes_on_device=True, shape=(D, H, W))
out = torch.empty((H, W), device=dev, dtype=torch.float32)
rng = np.random.default_rng(7)
for unit, slots in zip((15, 45, 135), (9, 3, 3)):
    units_, shared, filt, sets = pm.layer_geometry(W, H, 20, unit)
    batches, evals, wgs = [], 0, []
    for cells in sets:
        b = api.Batch(e, filt[cells], shared[cells])
        pl = np.zeros((len(cells), 4), np.float32)
        pl[:, 0] = rng.uniform(-0.05, 0.05, len(cells)); pl[:, 1] = rng.uniform(-0.05, 0.05, len(cells))
        cx, cy = shared[cells]["x"] + shared[cells]["w"] / 2, shared[cells]["y"] + shared[cells]["h"] / 2
        pl[:, 2] = rng.uniform(0.2, 0.8, len(cells)) * (D - 1) - pl[:, 0] * cx - pl[:, 1] * cy
        batches.append((b, torch.from_numpy(pl).to(dev)))
        evals += slots * int(sum(int(f["w"]) * int(f["h"]) for f in filt[cells]))
        wgs.append(b.num_jobs)
    def step():
        for b, p in batches:
            for _ in range(slots):
                b.run(p.data_ptr(), out.data_ptr(), mode=0, check=True, planes_on_device=True)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"unit {unit}: {len(batches) * slots} launches, workgroups per launch {min(wgs)}..{max(wgs)}, kernel kind {batches[0][0].kernel_kind(0)}, {ms:.2f} ms, "
          f"{evals / ms / 1e6:.1f} G filter-domain evals/s, {ms / (len(batches) * slots) * 1e3:.0f} us per launch")
