#!/usr/bin/env python
"""PCIe-inclusive rate of the boundary when it hands over HOST buffers (les_hip_unary_batch: host planes in,
host cost map out), for the note in DESIGN.md -- this is never bench.py's `value`.
Workload: n whole-image planes of the 1500x1000x256 configuration through the host-buffer entry point."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from localexpstereo_amd import api, synth  # noqa: E402

H, W, D, n = 1000, 1500, 256, 32
guide = synth.make_guide(H, W, 1234)
vol = torch.rand((D, H, W), device="cuda", dtype=torch.float32)
e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, volumes_on_device=True, shape=(D, H, W))
full = [(0, 0, W, H)] * n
planes = synth.fronto_planes(D)[:n]
cm = np.zeros((H, W), np.float32)
e.unary_batch(full[:2], full[:2], planes[:2], cm, check=False)
t0 = time.perf_counter()
e.unary_batch(full, full, planes, cm, check=False)          # n kernels' worth of work + n D2H copies of H*W floats
t1 = time.perf_counter()
print(json.dumps({"planes": n, "seconds": round(t1 - t0, 4), "Mcost_evals_per_s_pcie_inclusive": round(n * H * W / (t1 - t0) / 1e6, 1),
                  "d2h_bytes": n * H * W * 4}))
