"""Parity cases shared by the simulator tests (-m "not gpu") and the MI355X tests (-m gpu).

Every case runs the C-ABI library (`lib` = path of the .so: the HIP build on the GPU box, the CPU
SIMT-simulator build of the very same sources in the build container) and compares with the CPU
oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): aggregated costs within 1e-4 relative; written as
|got - ref| <= RTOL*|ref| + ATOL with ATOL = 1e-6 (costs live in [0, th_col=0.5]; the float32 ulp at
0.5 is 6e-8).  Invalid-label sentinels (1e6) and the set of written pixels must match exactly.
"""
import numpy as np

from localexpstereo_amd import api, synth
from oracle import oracle as om
from tests.util import load_cones_crop

RTOL = 1e-4
ATOL = 1e-6
TIGHT = 2e-6      # what the fp64-accumulating kernels actually achieve (absolute); regression guard


def compare_maps(got, ref, tight=True):
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "set of written pixels differs"
    m = ~np.isnan(ref)
    assert np.array_equal(got[m] == np.float32(1e6), ref[m] == np.float32(1e6)), "1e6 sentinels differ"
    v = m & (ref != np.float32(1e6))
    if not v.any():
        return 0.0
    err = np.abs(got[v].astype(np.float64) - ref[v])
    assert np.all(err <= RTOL * np.abs(ref[v]) + ATOL), f"parity: max abs err {err.max():.3e}"
    if tight:
        assert err.max() <= TIGHT, f"accuracy regression: max abs err {err.max():.3e}"
    return float(err.max())


class Pair:
    """An oracle context and a library context over the same inputs."""

    def __init__(self, lib, imL, imR, volL, volR, windR=20, eps=1e-4, th_col=0.5, max_disp=None, min_disp=0.0):
        self.o = om.Oracle(imL, imR, volL, volR, windR=windR, eps=eps, th_col=th_col, max_disp=max_disp, min_disp=min_disp)
        self.e = api.HipCostVolumeEnergy(imL, imR, volL, volR, windR=windR, eps=eps, th_col=th_col, max_disp=max_disp,
                                         min_disp=min_disp, lib=lib)
        self.H, self.W, self.D = self.o.H, self.o.W, self.o.D

    def close(self):
        self.e.close()


def cones_pair(lib, D=16, **kw):
    imL, imR = load_cones_crop()
    H, W = imL.shape[:2]
    return Pair(lib, imL, imR, synth.make_volume(D, H, W, 42), synth.make_volume(D, H, W, 43), **kw)


def synth_pair(lib, H, W, D, **kw):
    return Pair(lib, synth.make_guide(H, W, 1234), synth.make_guide(H, W, 1235), synth.make_volume(D, H, W, 42),
                synth.make_volume(D, H, W, 43), **kw)


def random_planes(n, D, H, W, seed, slant=0.3):
    rng = np.random.default_rng(seed)
    p = np.zeros((n, 4), np.float32)
    p[:, 0] = rng.uniform(-slant, slant, n)
    p[:, 1] = rng.uniform(-slant, slant, n)
    zc = rng.uniform(-2, D + 1, n)
    p[:, 2] = zc - p[:, 0] * rng.uniform(0, W, n) - p[:, 1] * rng.uniform(0, H, n)
    return p


# ------------------------------------------------------------------------------------------------
def case_stats(pr):
    """Guide statistics as consumed by the kernels vs LES/GuidedFilter.h:58-102 in double."""
    for mode in (0, 1):
        st = pr.e.stats(mode).astype(np.float64)
        so = pr.o.stats(mode)
        for k in range(3):
            assert np.max(np.abs(st[..., k, 0] - (so[3 + k] - 0.5))) <= 6e-8
        idx = {(0, 0): 6, (0, 1): 7, (0, 2): 8, (1, 1): 9, (1, 2): 10, (2, 2): 11}
        for (a, b), j in idx.items():
            for (r, c) in ((a, b), (b, a)):
                ref = so[j]
                assert np.max(np.abs(st[..., r, 1 + c] - ref) / np.maximum(np.abs(ref), 1e-3)) <= 2e-7


def case_single_calls(pr):
    """ComputeUnaryPotential for individual (filterRect, targetRect, plane, mode) calls, including
    image-corner cells, a whole-image call, an almost entirely invalid label and a 1x1 target."""
    H, W, D = pr.H, pr.W, pr.D
    calls = [
        (0, (0, 0, 62, 62), (0, 0, 42, 42), (0.0, 0.0, 3.0, 0.0)),
        (0, (19, 22, 82, 74), (39, 42, 42, 34), (0.05, -0.03, 4.25, 0.0)),
        (1, (W - 62, H - 62, 62, 62), (W - 42, H - 42, 42, 42), (-0.11, 0.07, 9.5, 0.0)),
        (0, (0, 0, W, H), (0, 0, W, H), (0.01, 0.02, 2.125, 0.0)),
        (1, (30, 0, 90, 60), (50, 0, 50, 40), (0.3, 0.2, -20.0, 0.0)),
        (0, (0, 36, 70, 60), (0, 56, 50, 40), (0.0, 0.0, float(D - 1), 0.0)),
        (0, (40, 40, 41, 41), (60, 60, 1, 1), (0.02, 0.01, 5.0, 0.0)),
        (1, (5, 5, 30, 30), (5, 5, 30, 30), (0.0, 0.0, 2.5, 0.0)),          # target == filter (no margin)
        (0, (10, 10, 50, 3), (12, 11, 40, 1), (0.0, 0.1, 1.0, 0.0)),        # degenerate thin rects
    ]
    worst = 0.0
    for mode, fr, tr, pl in calls:
        for check in (True, False):
            ref = pr.o.unary(fr, tr, pl, mode=mode, check=check)
            got = pr.e.ComputeUnaryPotential(fr, tr, np.full((H, W), np.nan, np.float32), pl, mode=mode, check=check)
            worst = max(worst, compare_maps(got, ref))
    return worst


def case_special_planes(pr):
    """NaN / inf planes and clamping (LES/CostVolumeEnergy.h:78-96)."""
    H, W = pr.H, pr.W
    fr, tr = (10, 8, 80, 70), (30, 28, 40, 30)
    for pl in [(float("nan"), 0.0, 1.0, 0.0), (0.0, 0.0, float("inf"), 0.0), (0.0, 0.0, float("-inf"), 0.0),
               (0.0, 0.0, -5.0, 0.0), (0.0, 0.0, 1e9, 0.0), (2.0, -3.0, 7.0, 0.0)]:
        ref = pr.o.unary(fr, tr, pl, check=False)
        got = pr.e.ComputeUnaryPotentialWithoutCheck(fr, tr, np.full((H, W), np.nan, np.float32), pl)
        compare_maps(got, ref)
        ref = pr.o.unary(fr, tr, pl, check=True)
        got = pr.e.ComputeUnaryPotential(fr, tr, np.full((H, W), np.nan, np.float32), pl)
        compare_maps(got, ref)


def case_cell_batches(pr, unit, sets=(0, 5, 15), seed=3, mode=0):
    """One lock-step of a disjoint set of LayerManager cells (LES/FastGCStereo.h:30-49)."""
    layer = om.Layer(pr.W, pr.H, 20, unit)
    worst = 0.0
    for s in sets:
        if s >= len(layer.sets):
            continue
        cells = layer.sets[s]
        planes = random_planes(len(cells), pr.D, pr.H, pr.W, seed + s)
        frs, trs = layer.filter[cells], layer.shared[cells]
        ref = pr.o.unary_batch(frs, trs, planes, mode=mode, check=True)
        got = pr.e.unary_batch(frs, trs, planes, mode=mode, check=True)
        worst = max(worst, compare_maps(got, ref))
    return worst


def case_init_cells(pr, unit=14, seed=9):
    """initCurrentFast geometry: filter = unit +- windR, target = unit (LES/FastGCStereo.h:105-114)."""
    layer = om.Layer(pr.W, pr.H, 20, unit)
    n = len(layer.unit)
    frs = np.zeros(n, api.RECT_DT)
    for i, u in enumerate(layer.unit):
        x0, y0 = max(0, u["x"] - 20), max(0, u["y"] - 20)
        x1, y1 = min(pr.W, u["x"] + u["w"] + 20), min(pr.H, u["y"] + u["h"] + 20)
        frs[i] = (x0, y0, x1 - x0, y1 - y0)
    planes = random_planes(n, pr.D, pr.H, pr.W, seed, slant=0.1)
    ref = pr.o.unary_batch(frs, layer.unit, planes, check=True)
    got = pr.e.unary_batch(frs, layer.unit, planes, check=True)
    assert not np.isnan(ref).any()        # unit regions tile the image
    return compare_maps(got, ref)


def case_empty_and_errors(pr):
    got = pr.e.unary_batch(np.zeros(0, api.RECT_DT), np.zeros(0, api.RECT_DT), np.zeros((0, 4), np.float32))
    assert np.isnan(got).all()
    # empty target rect: nothing written, no error
    got = pr.e.unary_batch([(0, 0, 50, 50)], [(10, 10, 0, 0)], [(0, 0, 1, 0)])
    assert np.isnan(got).all()
    import pytest
    with pytest.raises(api.LesHipError):
        pr.e.unary_batch([(0, 0, 50, 50)], [(40, 40, 20, 20)], [(0, 0, 1, 0)])       # target outside filter
    with pytest.raises(api.LesHipError):
        pr.e.unary_batch([(-5, 0, 50, 50)], [(0, 0, 20, 20)], [(0, 0, 1, 0)])        # filter outside image


def run_slabs(pr, planes, mode=0, check=False):
    """Whole-image aggregation of n hypothesis planes into [n][H][W]."""
    n = len(planes)
    full = [(0, 0, pr.W, pr.H)] * n
    b = api.Batch(pr.e, full, full, out_slabs=True)
    buf = api.DeviceBuffer(pr.e, n * pr.H * pr.W * 4)
    buf.fill(0xFF)
    b.run(planes, buf.ptr, mode=mode, check=check)
    pr.e.synchronize()
    out = buf.download((n, pr.H, pr.W), np.float32)
    buf.free()
    b.destroy()
    return out


def case_plane_slabs(pr, n=5, mode=1):
    planes = np.concatenate([synth.fronto_planes(pr.D)[:2], random_planes(n - 2, pr.D, pr.H, pr.W, 21, slant=0.2)])
    out = run_slabs(pr, planes, mode=mode, check=False)
    worst = 0.0
    for i in range(n):
        ref = pr.o.unary((0, 0, pr.W, pr.H), (0, 0, pr.W, pr.H), tuple(planes[i]), mode=mode, check=False)
        worst = max(worst, compare_maps(out[i], ref))
    return worst


def case_wta(pr, seed=5):
    """Device WTA update vs LES/FastGCStereo.h:56-60."""
    import ctypes as C
    H, W = pr.H, pr.W
    rng = np.random.default_rng(seed)
    cur = rng.random((H, W), dtype=np.float32)
    prop = rng.random((H, W), dtype=np.float32)
    prop[::7, ::5] = cur[::7, ::5]                     # ties: strict '>' keeps the current label
    labels = np.zeros((H, W), api.PLANE_DT)
    labels["c"] = -1.0
    layer = om.Layer(W, H, 20, 14)
    cells = layer.sets[3]
    rects = layer.shared[cells]
    planes = random_planes(len(cells), pr.D, H, W, seed)
    dc, dp, dl = (api.DeviceBuffer(pr.e, a.nbytes) for a in (cur, prop, labels))
    dc.upload(cur); dp.upload(prop); dl.upload(labels)
    pr.e.wta_update(rects, planes, dc.ptr, dp.ptr, dl.ptr)
    pr.e.synchronize()
    gc, gl = dc.download((H, W), np.float32), dl.download((H, W), api.PLANE_DT)
    L = om.lib()
    rc, rl = cur.copy(), labels.copy()
    for r, p in zip(rects, planes):
        L.les_oracle_wta_update(W, om.Rect(*[int(v) for v in r]), rc.ctypes.data_as(C.c_void_p), prop.ctypes.data_as(C.c_void_p),
                                rl.ctypes.data_as(C.c_void_p), om.Plane(*[float(v) for v in p]))
    assert np.array_equal(gc, rc)
    assert gl.tobytes() == rl.tobytes()
    for d in (dc, dp, dl):
        d.free()
