"""Do two concurrent lock-steps of small cuts (two views, two host threads, one OpenMP team each) slow each other down on this host?
Synthetic layer-0 shaped lock-steps (1400 cells of 42 x 42 nodes) through les_gc_solve_prebuilt, alone and two at a time."""
import os, sys, time, threading
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.getcwd())
import numpy as np
from localexpstereo_amd import gc as lgc, api
n, w = 1400, 42
rng = np.random.default_rng(1)
reg = np.zeros(n, dtype=api.RECT_DTYPE if hasattr(api, "RECT_DTYPE") else [("x", "i4"), ("y", "i4"), ("w", "i4"), ("h", "i4")])
reg["w"] = w; reg["h"] = w
off = (np.arange(n, dtype=np.int64) * w * w)
def make():
    p = rng.uniform(0, 0.5, (n * w * w, 5)).astype(np.float32)
    p[:, 0] = rng.normal(0, 0.7, n * w * w)
    q = p.reshape(n, w, w, 5)                      # arcs E, S, SW, SE that would leave the region carry no capacity
    q[:, :, -1, 1] = 0; q[:, -1, :, 2] = 0; q[:, -1, :, 3] = 0; q[:, :, 0, 3] = 0; q[:, -1, :, 4] = 0; q[:, :, -1, 4] = 0
    return p.reshape(-1).copy(), np.zeros(n * w * w, np.uint8)
A, B = make(), make()
def loop(pay, masks, reps, nt, out):
    t = time.perf_counter()
    for _ in range(reps):
        lgc.solve_prebuilt(reg, pay, off, masks, nthreads=nt)
    out.append((time.perf_counter() - t) / reps)
for nt in (8, 16, 24):
    o = []
    loop(*A, 20, nt, o)
    alone = o[0]
    o = []
    ths = [threading.Thread(target=loop, args=(*X, 20, nt, o)) for X in (A, B)]
    [t.start() for t in ths]; [t.join() for t in ths]
    print(f"threads per team {nt}: one lock-step alone {alone * 1e3:.2f} ms; two concurrent teams {o[0] * 1e3:.2f} / {o[1] * 1e3:.2f} ms each")
