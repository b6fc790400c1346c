import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from localexpstereo_amd import synth
from oracle import oracle as om
H, W, D = 1000, 1500, 256
guide = synth.make_guide(H, W, 1234)
ns = 96
vol = np.random.default_rng(0).random((ns + 1, H, W), dtype=np.float32)
o = om.Oracle(guide, None, vol, None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1)
planes = synth.fronto_planes(D)
for nt in (256, 16, 24, 32, 64, 128):
    o.aggregate_planes(planes[:min(ns, nt)], nthreads=nt)
    t = time.perf_counter(); o.aggregate_planes(planes[:ns], nthreads=nt); dt = time.perf_counter() - t
    print(nt, "threads:", round(ns * H * W / dt / 1e6, 1), "Mevals/s")
