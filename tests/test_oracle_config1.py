"""BASELINE configs[0]: MiddV2 cones 450x375, ndisp 64, pmIterations = 2, CPU reference path (no GPU).
The oracle's NaiveStereoEnergy restatement (LES/StereoEnergy.h:629-764) driven by the reference's own loop order
(LES/FastGCStereo.h:22-72,94-115, doGC == false) on the reference's bundled cones pair, evaluated with the
Evaluator's bad-pixel definition (LES/Evaluator.h:133-140, threshold 0.5 px, LES/main.cpp:280).
The data set lives under /root/reference (not available on the GPU box): the test skips when it is absent."""
import os
import time

import numpy as np
import pytest

CONES = "/root/reference/data/MiddV2/cones"


@pytest.mark.skipif(not os.path.isdir(CONES), reason="reference data not present")
def test_config1_cones_cpu_path(oracle_mod):
    from PIL import Image
    om = oracle_mod
    load = lambda n: np.ascontiguousarray(np.asarray(Image.open(os.path.join(CONES, n)).convert("RGB"))[:, :, ::-1])
    imL, imR = load("imL.png"), load("imR.png")
    gt = np.asarray(Image.open(os.path.join(CONES, "groundtruth.png"))).astype(np.float32) / 4.0     # info.txt: scale 4
    nonocc = np.asarray(Image.open(os.path.join(CONES, "nonocc.png")).convert("L")) == 255             # LES/main.cpp:263
    H, W = imL.shape[:2]
    assert (W, H) == (450, 375)
    maxd = 63.0                                                                                       # -ndisp 64
    o = om.Oracle.naive(imL, imR, maxd, windR=20, eps=1e-4, alpha=0.9, th_col=10.0, th_grad=2.0)
    layers = [om.Layer(W, H, 20, u) for u in (5, 15, 25)]                                             # LES/main.cpp:300-306
    tables = [[(0, 1), (2, 1), (1, 7)], [(0, 2), (2, 1)], [(0, 2), (2, 1)]]
    rng = np.random.default_rng(0)
    states = [rng.integers(1, 2**63, len(L.unit), dtype=np.uint64) for L in layers]
    labels = np.zeros((H, W), om.PLANE_DT)
    cur = np.zeros((H, W), np.float32)
    prop = np.zeros((H, W), np.float32)
    known = gt > 0

    def bad(mask, thr=0.5):
        ys, xs = np.mgrid[0:H, 0:W]
        d = labels["a"] * xs + labels["b"] * ys + labels["c"]
        return float((np.abs(d - gt)[mask] > thr).mean() * 100)

    t0 = time.time()
    o.pm_init(layers[0].unit, states[0].copy(), labels, cur)
    hist = [(bad(known), bad(known & nonocc), float(cur.sum()))]
    for it in range(2):                                                                               # -pmIterations 2
        for L, tab, st in zip(layers, tables, states):
            for cells in L.sets:
                s = np.ascontiguousarray(st[cells])
                o.pm_set(L.unit[cells], L.shared[cells], L.filter[cells], s, tab, labels, cur, prop, it)
                st[cells] = s
        hist.append((bad(known), bad(known & nonocc), float(cur.sum())))
    print("config 1 (cones, CPU path): bad0.5 all / nonocc / energy per iteration:", hist, f"{time.time() - t0:.1f} s")
    assert hist[0][0] > 90.0
    assert hist[1][2] < hist[0][2] and hist[2][2] <= hist[1][2] + 1e-3
    assert hist[-1][1] < 30.0          # PatchMatch iterations alone (no graph cut, no smoothness term) already find most surfaces
