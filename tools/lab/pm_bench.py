#!/usr/bin/env python
"""Times the device-resident PatchMatch iterations (initCurrentFast + pmIterations, doGC = false) at a
MiddV3-like size with the reference's layer set-up (LES/main.cpp:391-397): unit sizes int(w*.01/.03/.09),
proposers {Exp(1),Ransac(1),Random(7)}, {Exp(2),Ransac(1)}, {Exp(2),Ransac(1)}.  Synthetic data.

usage: python tools/pm_bench.py [W H D pm_iterations]     (default: Adirondack-H shape 1436 992 256 2)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from localexpstereo_amd import api, pm, synth  # noqa: E402


def main():
    W, H, D, iters = (int(v) for v in (sys.argv[1:5] + [1436, 992, 256, 2][len(sys.argv) - 1:]))
    dev = torch.device("cuda", 0)
    guide = synth.make_guide(H, W, 1234)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    vol = torch.rand((D, H, W), device=dev, dtype=torch.float32, generator=gen)
    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1,
                                volumes_on_device=True, shape=(D, H, W))
    units = (int(W * 0.01), int(W * 0.03), int(W * 0.09))
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)],
             [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
    r = pm.PMRunner(e, units, table, seed=3, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.init_labels()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    per_iter = []
    for it in range(iters):
        a = time.perf_counter()
        r.iteration(it)
        torch.cuda.synchronize()
        per_iter.append(time.perf_counter() - a)
    cur = r.cur.cpu().numpy()
    evals = 0
    for li, layer in enumerate(r.shards):
        k = sum(K for _, K in table[li])
        for sh in layer:
            evals += k * int(sum(int(f["w"]) * int(f["h"]) for f in sh.batch.frs))
    print(json.dumps({"shape": [W, H, D], "layer_units": units, "init_s": round(t1 - t0, 4), "iteration_s": [round(x, 4) for x in per_iter],
                      "filter_domain_evals_per_iteration": evals,
                      "Mcost_evals_per_s_filter_domain": round(evals / per_iter[-1] / 1e6, 1),
                      "mean_cost": float(cur.mean()), "frac_valid": float((cur < 1e5).mean())}))


if __name__ == "__main__":
    main()
