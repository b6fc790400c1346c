"""Parity cases shared by the simulator tests (-m "not gpu") and the MI355X tests (-m gpu).

Every case runs the C-ABI library (`lib` = path of the .so: the HIP build on the GPU box, the CPU
SIMT-simulator build of the very same sources in the build container) and compares with the CPU
oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): aggregated costs within 1e-4 relative; written as
|got - ref| <= RTOL*|ref| + ATOL with ATOL = 1e-6 (costs live in [0, th_col=0.5]; the float32 ulp at
0.5 is 6e-8).  Invalid-label sentinels (1e6) and the set of written pixels must match exactly.
"""
import os

import numpy as np

from localexpstereo_amd import api, synth
from oracle import oracle as om
from tests.util import GOLDEN, load_cones_crop

RTOL = 1e-4
ATOL = 1e-6
TIGHT = 2e-6      # what the kernels actually achieve (absolute, costs in [0, 0.5]); regression guard
NAIVE_TIGHT = 2e-5  # the image-based matching cost (config 1) lives in [0, 2.8] (th_col 10, th_grad 2, alpha 0.9): achieved 1.1e-5 absolute = 4e-6 of the range


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
        bound = TIGHT if tight is True else float(tight)
        assert err.max() <= bound, f"accuracy regression: max abs err {err.max():.3e} > {bound:.1e}"
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


class NaivePair:
    """Oracle and library contexts of the image-based matching cost (NaiveStereoEnergy, LES/StereoEnergy.h:629-764)."""

    def __init__(self, lib, imL, imR, max_disp, windR=20, eps=1e-4, alpha=0.9, th_col=10.0, th_grad=2.0):
        self.o = om.Oracle.naive(imL, imR, max_disp, windR=windR, eps=eps, alpha=alpha, th_col=th_col, th_grad=th_grad)
        self.e = api.HipCostVolumeEnergy.naive(imL, imR, windR=windR, eps=eps, alpha=alpha, th_col=th_col, th_grad=th_grad,
                                               max_disp=max_disp, lib=lib)
        self.H, self.W, self.D = self.o.H, self.o.W, int(max_disp) + 1

    def close(self):
        self.e.close()


def case_naive(lib, windR=20, pm=True):
    """Config 1's energy (MiddV2 parameters, LES/main.cpp:86-121) on the cones crop: single calls in both views with
    valid / partly invalid / out-of-image / NaN planes, a lock-step of cells, and one PatchMatch set update."""
    imL, imR = load_cones_crop()
    pr = NaivePair(lib, imL, imR, 31.0, windR=windR)
    H, W = pr.H, pr.W
    R2 = windR
    worst = 0.0
    calls = [
        (0, (0, 0, 62, 62), (0, 0, 42, 42), (0.0, 0.0, 12.0, 0.0)),
        (0, (19, 22, 82, 74), (39, 42, 42, 34), (0.05, -0.03, 14.25, 0.0)),
        (1, (W - 62, H - 62, 62, 62), (W - 42, H - 42, 42, 42), (-0.11, 0.07, 9.5, 0.0)),
        (0, (0, 0, W, H), (0, 0, W, H), (0.01, 0.02, 10.125, 0.0)),
        (1, (0, 0, W, H), (0, 0, W, H), (-0.02, 0.01, 17.3, 0.0)),
        (1, (30, 0, 90, 60), (50, 0, 50, 40), (0.3, 0.2, -20.0, 0.0)),                 # warps far outside the other view
        (0, (0, 36, 70, 60), (0, 56, 50, 40), (0.0, 0.0, 31.0, 0.0)),
        (0, (40, 40, 41, 41), (60, 60, 1, 1), (0.02, 0.01, 5.0, 0.0)),
        (0, (10, 8, 80, 70), (30, 28, 40, 30), (float("nan"), 0.0, 1.0, 0.0)),
        (0, (10, 8, 80, 70), (30, 28, 40, 30), (0.0, 0.0, float("inf"), 0.0)),
        (1, (10, 8, 80, 70), (30, 28, 40, 30), (0.0, 0.0, 1e9, 0.0)),
    ]
    for mode, fr, tr, pl in calls:
        for check in (True, False):
            ref = pr.o.unary(fr, tr, pl, mode=mode, check=check)
            got = pr.e.ComputeUnaryPotential(fr, tr, np.full((H, W), np.nan, np.float32), pl, mode=mode, check=check)
            worst = max(worst, compare_maps(got, ref, tight=NAIVE_TIGHT))
    layer = om.Layer(W, H, windR, 9)
    for s, mode in ((0, 0), (7, 1)):
        cells = layer.sets[s]
        planes = random_planes(len(cells), 32, H, W, 17 + s, slant=0.2)
        ref = pr.o.unary_batch(layer.filter[cells], layer.shared[cells], planes, mode=mode, check=True)
        got = pr.e.unary_batch(layer.filter[cells], layer.shared[cells], planes, mode=mode, check=True)
        worst = max(worst, compare_maps(got, ref, tight=NAIVE_TIGHT))
        # which kernel served the lock-step: the march kernel (raw-cost patches + role A's one-tap path) where the radius has an
        # instantiation and LES_HIP_KERNEL does not force the strip kernel; both must meet the same bound
        bt = api.Batch(pr.e, layer.filter[cells], layer.shared[cells])
        want = 1 if (2 <= windR // 2 <= 10 and os.environ.get("LES_HIP_KERNEL", "") != "strip") else 0
        assert bt.kernel_kind(mode) == want, (windR, bt.kernel_kind(mode))
        bt.destroy()
    pr.close()
    return worst


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


def case_widened_abi_errors(lib):
    """Error behaviour of the entry points added for the "next" rows: status codes with a message, never a crash or a
    silent CPU path."""
    import pytest
    imL, imR = load_cones_crop()
    H, W = imL.shape[:2]
    vol = synth.make_volume(8, H, W, 1)
    with pytest.raises(api.LesHipError, match="both views"):
        _naive_one_view(imL, lib)
    e = api.HipCostVolumeEnergy(imL, None, vol, None, lib=lib)                      # left view only
    lab = api.DeviceBuffer(e, H * W * 16)
    lab.fill(0)
    with pytest.raises(api.LesHipError, match="both views"):
        e.post_process(lab.ptr, lab.ptr)
    with pytest.raises(api.LesHipError, match="null"):
        e.post_process(0, lab.ptr)
    with pytest.raises(api.LesHipError):
        e.consistency_check(lab.ptr, lab.ptr, 0, 0)
    b = api.Batch(e, [(0, 0, 40, 40)], [(10, 10, 20, 20)])
    assert b.graph_nodes() == 400 and b.graph_offsets().tolist() == [0]
    with pytest.raises(api.LesHipError, match="view 1"):
        b.expansion_graph(lab.ptr, lab.ptr, lab.ptr, lab.ptr, lab.ptr, mode=1)
    with pytest.raises(api.LesHipError, match="null"):
        b.expansion_graph(lab.ptr, lab.ptr, 0, lab.ptr, lab.ptr)
    with pytest.raises(api.LesHipError, match="null"):
        b.apply_masks(lab.ptr, 0, lab.ptr, lab.ptr, lab.ptr)
    b.destroy(); lab.free(); e.close()


def _naive_one_view(imL, lib):
    L = api.load(lib)
    import ctypes as C
    p = api.Params(imL.shape[0], imL.shape[1], 1, 20, 1e-4, 10.0, 15.0, 0.0, 0, 0)
    h = C.c_void_p()
    rc = L.les_hip_create_naive(C.byref(h), C.byref(p), api._ptr(np.ascontiguousarray(imL)), None, C.c_float(0.9), C.c_float(2.0))
    if rc:
        raise api.LesHipError(L.les_hip_last_error().decode())
    L.les_hip_destroy(h)


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


def case_tiled_taps(lib, H=150, W=203, D=24, monkeypatch=None):
    """Steep planes take their two taps from the tiled copy of the volume ([H][W/8][D][8], csrc/les_march.h role A KIND 5), all others from
    [D][H][W]: the same values at other addresses, so a context with the copy (default) and one without (LES_HIP_TILED=0) agree BIT FOR BIT,
    and both agree with the oracle.  W is not a multiple of the tile width; planes leave the disparity range on both sides (clamped taps),
    slopes on both sides of the thresholds, both views, whole-image slabs (wide jobs) and layer-0 cells (two narrow jobs per workgroup)."""
    rng = np.random.default_rng(5)
    n = 10
    planes = np.zeros((n, 4), np.float32)
    planes[:, 0] = [0.5, -0.5, 0.3, -0.26, 0.13, -0.12, 0.06, -0.04, 0.9, -1.7]
    planes[:, 1] = rng.uniform(-0.2, 0.2, n)
    planes[:, 2] = rng.uniform(0.2, 0.8, n) * (D - 1) - planes[:, 0] * W / 2 - planes[:, 1] * H / 2
    outs, worst = [], 0.0
    for tiled in ("1", "0"):
        monkeypatch.setenv("LES_HIP_TILED", tiled)            # read when a context builds its march tables
        pr = synth_pair(lib, H, W, D)
        try:
            assert [pr.e.tiled_volume_bytes(m) for m in (0, 1)] == ([H * ((W + 7) // 8) * 8 * D * 4] * 2 if tiled == "1" else [0, 0])
            res = [run_slabs(pr, planes, mode=m, check=True) for m in (0, 1)]
            layer = om.Layer(pr.W, pr.H, 20, 15)
            cells = layer.sets[3]
            cp = np.tile(planes, (len(cells) // n + 1, 1))[:len(cells)]
            res.append(pr.e.unary_batch(layer.filter[cells], layer.shared[cells], cp, mode=0, check=True))
            if tiled == "1":
                ref = pr.o.unary_batch(layer.filter[cells], layer.shared[cells], cp, mode=0, check=True)
                worst = max(worst, compare_maps(res[2], ref))
                for m in (0, 1):
                    for i in range(n):
                        ref = pr.o.unary((0, 0, pr.W, pr.H), (0, 0, pr.W, pr.H), tuple(planes[i]), mode=m, check=True)
                        worst = max(worst, compare_maps(res[m][i], ref))
            outs.append(res)
        finally:
            pr.close()
    for m in (0, 1, 2):
        assert np.array_equal(outs[0][m].view(np.uint32), outs[1][m].view(np.uint32)), "tiled and planar taps must give identical costs"
    return worst


def case_grouped_slots(pr, unit=14, set_index=3, slots=3, mode=0):
    """Several proposal slots of one disjoint set in ONE launch (les_hip_batch_create with out_slabs = cells per slot): slot s of every cell
    into cost map s.  Every map must equal the single-slot lock-step of the oracle, written pixels and sentinels included."""
    layer = om.Layer(pr.W, pr.H, 20, unit)
    cells = layer.sets[min(set_index, len(layer.sets) - 1)]
    n = len(cells)
    frs, trs = np.tile(layer.filter[cells], slots), np.tile(layer.shared[cells], slots)
    planes = random_planes(n * slots, pr.D, pr.H, pr.W, 31, slant=0.1)
    b = api.Batch(pr.e, frs, trs, out_slabs=n)
    buf = api.DeviceBuffer(pr.e, slots * pr.H * pr.W * 4)
    buf.fill(0xFF)                                      # NaN pattern: unwritten pixels stay NaN like the oracle's fresh map
    b.run(planes, buf.ptr, mode=mode, check=True)
    pr.e.synchronize()
    out = buf.download((slots, pr.H, pr.W), np.float32)
    buf.free()
    kind = b.kernel_kind(mode)
    b.destroy()
    worst = 0.0
    for s_ in range(slots):
        ref = pr.o.unary_batch(layer.filter[cells], layer.shared[cells], planes[s_ * n:(s_ + 1) * n], mode=mode, check=True)
        worst = max(worst, compare_maps(out[s_], ref))
    return worst, kind


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


# ------------------------------------------------------------------------------------------------
# hypothesis generation (LES/Proposer.h, LES/StereoEnergy.h:120-129) and the PatchMatch lock-step
# ------------------------------------------------------------------------------------------------
def _seeds(n, seed):
    """Non-zero 64-bit generator states, one per cell."""
    x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(31)
    return np.where(x == 0, np.uint64(0xFFFFFFFF), x).astype(np.uint64)


def _label_map(H, W, D, seed, noise=0.0, block=(9, 11)):
    rng = np.random.default_rng(seed)
    lab = np.zeros((H, W), api.PLANE_DT)
    ys, xs = np.mgrid[0:H, 0:W]
    by, bx = ys // block[0], xs // block[1]
    na, nb = by.max() + 1, bx.max() + 1
    A = rng.uniform(-0.2, 0.2, (na, nb)).astype(np.float32)
    B = rng.uniform(-0.2, 0.2, (na, nb)).astype(np.float32)
    Z = rng.uniform(1, D - 2, (na, nb)).astype(np.float32)
    lab["a"], lab["b"] = A[by, bx], B[by, bx]
    lab["c"] = Z[by, bx] - lab["a"] * xs.astype(np.float32) - lab["b"] * ys.astype(np.float32)
    if noise:
        lab["c"] += rng.normal(0, noise, (H, W)).astype(np.float32)
    return lab


def _oracle_proposals(kind, labels, W, units, seeds, m, mind, maxd):
    import ctypes as C
    L = om.lib()
    out = np.zeros(len(units), api.PLANE_DT)
    states = np.zeros(len(units), np.uint64)
    lp = labels.ctypes.data_as(C.c_void_p)
    for i, u in enumerate(units):
        r = om.Rng(int(seeds[i]))
        rect = om.Rect(int(u["x"]), int(u["y"]), int(u["w"]), int(u["h"]))
        if kind == api.PROPOSE_EXPANSION:
            p = L.les_expansion_proposal(C.byref(r), lp, W, rect)
        elif kind == api.PROPOSE_RANDOM:
            p = L.les_random_proposal(C.byref(r), lp, W, rect, m, mind, maxd)
        elif kind == api.PROPOSE_RANSAC:
            p = L.les_ransac_proposal(C.byref(r), lp, W, rect, 500, 0.95, 1.0)
        else:
            px, py = C.c_int(), C.c_int()
            L.les_select_random_pixel(C.byref(r), rect, C.byref(px), C.byref(py))
            p = L.les_create_random_label(C.byref(r), mind, maxd, px.value, py.value)
        out[i] = (p.a, p.b, p.c, p.v)
        states[i] = r.state
    return out, states


def case_proposers(pr, unit=14, set_index=5, seed=11):
    H, W, D = pr.H, pr.W, pr.D
    mind, maxd = 0.0, float(D - 1)
    layer = om.Layer(W, H, 20, unit)
    cells = layer.sets[min(set_index, len(layer.sets) - 1)]
    units = layer.unit[cells]
    n = len(cells)
    b = api.Batch(pr.e, layer.filter[cells], layer.shared[cells])
    b.set_units(units)
    labels = _label_map(H, W, D, seed, noise=0.3)
    d_lab = api.DeviceBuffer(pr.e, labels.nbytes)
    d_rng = api.DeviceBuffer(pr.e, 8 * n)
    d_pl = api.DeviceBuffer(pr.e, 16 * n)
    try:
        for kind, m in ((api.PROPOSE_EXPANSION, 0), (api.PROPOSE_RANDOM, 0), (api.PROPOSE_RANDOM, 4), (api.PROPOSE_RANSAC, 0),
                        (api.PROPOSE_INIT, 0)):
            seeds = _seeds(n, seed + 7 * kind + m)
            d_lab.upload(labels)
            d_rng.upload(seeds)
            b.propose(kind, d_lab.ptr, d_rng.ptr, d_pl.ptr, m=m)
            pr.e.synchronize()
            got = d_pl.download((n,), api.PLANE_DT)
            st = d_rng.download((n,), np.uint64)
            ref, rst = _oracle_proposals(kind, labels, W, units, seeds, m, mind, maxd)
            g4 = got.view(np.float32).reshape(n, 4)
            r4 = ref.view(np.float32).reshape(n, 4)
            if kind == api.PROPOSE_EXPANSION:
                assert got.tobytes() == ref.tobytes() and np.array_equal(st, rst)
            elif kind in (api.PROPOSE_RANDOM, api.PROPOSE_INIT):
                assert np.array_equal(st, rst)
                # device vs host libm (cos/sin in double) may differ in the last bit after the cast to float
                np.testing.assert_allclose(g4, r4, rtol=2e-6, atol=2e-6)
                assert np.mean(np.all(g4 == r4, axis=1)) > 0.8
                if kind == api.PROPOSE_INIT:
                    lab_after = d_lab.download((H, W), api.PLANE_DT)
                    for i, u in enumerate(units):
                        blk = lab_after[u["y"]:u["y"] + u["h"], u["x"]:u["x"] + u["w"]]
                        assert np.all(blk == got[i])
            else:
                # RANSAC has no trigonometry.  The oracle accumulates the normal equations in the natural row order (round 4: no longer
                # shaped after the device's quad-interleaved order); both work in double and round once to float, so the planes agree to
                # float round-off and the sample / inlier decisions -- hence the generator states -- coincide.  Measured: bit-identical in
                # every cell tried (tools/ransac_agreement.py), which is a fact about the data, not part of the contract.
                assert np.array_equal(st, rst), f"ransac generator states differ in {int((st != rst).sum())} of {n} cells"
                np.testing.assert_allclose(g4, r4, rtol=1e-5, atol=1e-5)
                assert np.mean(np.all(g4 == r4, axis=1)) > 0.9
    finally:
        for d in (d_lab, d_rng, d_pl):
            d.free()
        b.destroy()


def case_ransac_schedule(pr, combos=((14, 0.0), (14, 0.3), (14, 3.0), (40, 1.0), (40, 30.0)), seed=23):
    """The device RANSAC follows the reference's ADAPTIVE schedule (LES/Proposer.h:193 `while (no_sam < max_sam)`, :229-236) in chunks of candidates
    (csrc/les_propose.h): label maps from exactly planar (the loop ends after its first sample) over noisy (a few dozen samples) to garbage (all 500)
    must give the oracle's planes and -- the proof that the same number of samples was consumed -- the oracle's generator state in EVERY cell.
    -> {(unit, noise): share of cells whose generator advanced past the first chunk of 16 candidates}"""
    H, W, D = pr.H, pr.W, pr.D
    mind, maxd = 0.0, float(D - 1)
    out = {}
    for unit, noise in combos:
        layer = om.Layer(W, H, 20, unit)
        cells = layer.sets[min(3, len(layer.sets) - 1)]
        units = layer.unit[cells]
        n = len(cells)
        b = api.Batch(pr.e, layer.filter[cells], layer.shared[cells])
        b.set_units(units)
        labels = _label_map(H, W, D, seed + unit, noise=noise, block=(H, W) if noise == 0.0 else (9, 11))     # noise 0: ONE plane over the image
        d_lab, d_rng, d_pl = api.DeviceBuffer(pr.e, labels.nbytes), api.DeviceBuffer(pr.e, 8 * n), api.DeviceBuffer(pr.e, 16 * n)
        try:
            seeds = _seeds(n, seed + 3 * unit)
            d_lab.upload(labels)
            d_rng.upload(seeds)
            b.propose(api.PROPOSE_RANSAC, d_lab.ptr, d_rng.ptr, d_pl.ptr, m=0)
            pr.e.synchronize()
            got = d_pl.download((n,), api.PLANE_DT).view(np.float32).reshape(n, 4)
            st = d_rng.download((n,), np.uint64)
            ref, rst = _oracle_proposals(api.PROPOSE_RANSAC, labels, W, units, seeds, 0, mind, maxd)
            ref = ref.view(np.float32).reshape(n, 4)
            assert np.array_equal(st, rst), f"unit {unit}, noise {noise}: generator states differ in {int((st != rst).sum())} of {n} cells"
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
            # how far the loop went: the first chunk is 16 candidates = at least 48 draws of the generator (duplicates inside a triple only add
            # draws); a state that is not among the first 48 + 8 successors of the seed belongs to a loop that went past the first chunk
            far = sum(_draws_between(int(seeds[i]), int(st[i]), 16 * 3 + 8) is None for i in range(n))
            out[(unit, noise)] = far / max(1, n)
        finally:
            for d in (d_lab, d_rng, d_pl):
                d.free()
            b.destroy()
    return out


def _draws_between(s0, s1, limit):
    """number of generator steps from state s0 to state s1 if at most `limit`, else None (cv::RNG: state = (uint32)state * 4164903690 + (state >> 32))"""
    s = s0
    for k in range(limit + 1):
        if s == s1:
            return k
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
    return None


def pm_iteration_oracle(pr, layers_units, proposer_table, seeds_per_layer, labels, cur, iteration, mode=0):
    """CPU PatchMatch iteration (doGC = false) with the oracle: LES/FastGCStereo.h:22-72 in lock-step
    order (all cells of a set draw proposal k, evaluate, WTA), which is equivalent to the reference's
    per-cell order because cells of a set are independent."""
    import ctypes as C
    L = om.lib()
    H, W, D = pr.H, pr.W, pr.D
    mind, maxd = 0.0, float(D - 1)
    for li, unit in enumerate(layers_units):
        layer = om.Layer(W, H, 20, unit)
        states = seeds_per_layer[li]
        for cells in layer.sets:
            units, shared, filt = layer.unit[cells], layer.shared[cells], layer.filter[cells]
            for kind, K in proposer_table[li]:
                it = 0
                while True:
                    if kind == api.PROPOSE_RANDOM:
                        if not L.les_random_is_continued(it, K, iteration, mind, maxd):
                            break
                    elif it >= K:
                        break
                    planes, new_states = _oracle_proposals(kind, labels, W, units, states[cells], iteration + it, mind, maxd)
                    states[cells] = new_states
                    prop = pr.o.unary_batch(filt, shared, planes, mode=mode, check=True)
                    for r, p in zip(shared, planes):
                        L.les_oracle_wta_update(W, om.Rect(*[int(v) for v in r]), cur.ctypes.data_as(C.c_void_p),
                                                prop.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p),
                                                om.Plane(*[float(v) for v in p]))
                    it += 1
    return labels, cur


def pm_iteration_device(pr, layers_units, proposer_table, seeds_per_layer, labels, cur, iteration, mode=0):
    """The same iteration with everything resident on the device (labels, costs, generator states)."""
    L = om.lib()
    H, W, D = pr.H, pr.W, pr.D
    mind, maxd = 0.0, float(D - 1)
    d_lab = api.DeviceBuffer(pr.e, labels.nbytes)
    d_cur = api.DeviceBuffer(pr.e, cur.nbytes)
    d_prop = api.DeviceBuffer(pr.e, cur.nbytes)
    d_lab.upload(labels)
    d_cur.upload(cur)
    for li, unit in enumerate(layers_units):
        layer = om.Layer(W, H, 20, unit)
        for cells in layer.sets:
            n = len(cells)
            b = api.Batch(pr.e, layer.filter[cells], layer.shared[cells])
            b.set_units(layer.unit[cells])
            d_rng = api.DeviceBuffer(pr.e, 8 * n)
            d_pl = api.DeviceBuffer(pr.e, 16 * n)
            d_rng.upload(np.ascontiguousarray(seeds_per_layer[li][cells]))
            for kind, K in proposer_table[li]:
                it = 0
                while True:
                    if kind == api.PROPOSE_RANDOM:
                        if not L.les_random_is_continued(it, K, iteration, mind, maxd):
                            break
                    elif it >= K:
                        break
                    b.propose(kind, d_lab.ptr, d_rng.ptr, d_pl.ptr, m=iteration + it)
                    b.run(d_pl.ptr, d_prop.ptr, mode=mode, check=True, planes_on_device=True)
                    b.wta(d_pl.ptr, d_cur.ptr, d_prop.ptr, d_lab.ptr)
                    it += 1
            pr.e.synchronize()
            seeds_per_layer[li][cells] = d_rng.download((n,), np.uint64)
            d_rng.free(); d_pl.free(); b.destroy()
    out_l, out_c = d_lab.download((H, W), api.PLANE_DT), d_cur.download((H, W), np.float32)
    for d in (d_lab, d_cur, d_prop):
        d.free()
    return out_l, out_c


def case_pm_iteration(pr, layers_units=(12, 36), seed=21, plane_exact=True):
    """One PatchMatch iteration (doGC=false): proposals -> unary costs -> winner-take-all
    (LES/FastGCStereo.h:41-61), device vs oracle.

    (a) lock-step check with the device state re-synchronised to the oracle's before every step: proposals
        equal (exactly in the simulator; to float round-off of device trig on the GPU), proposal costs within
        tolerance, and identical WTA decisions wherever the two costs differ by more than the float noise
        (a decision between two labels whose costs agree to 1e-5 is a tie at equal energy: either is valid).
    (b) free-running iteration on the device: the resulting total cost agrees with the oracle-driven run."""
    import ctypes as C
    L = om.lib()
    H, W, D = pr.H, pr.W, pr.D
    mind, maxd = 0.0, float(D - 1)
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)],     # LES/main.cpp:391-397
             [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
    labels0 = _label_map(H, W, D, seed, noise=0.2)
    cur0 = np.full((H, W), np.float32(1e6))
    seeds = [_seeds(len(om.Layer(W, H, 20, u).unit), seed + i) for i, u in enumerate(layers_units)]

    # ---- (a) teacher-forced lock-steps
    rl, rc = labels0.copy(), cur0.copy()
    d_lab, d_cur, d_prop = (api.DeviceBuffer(pr.e, a.nbytes) for a in (rl, rc, rc))
    steps = 0
    worst = 0.0
    for li, unit in enumerate(layers_units):
        layer = om.Layer(W, H, 20, unit)
        st = seeds[li].copy()
        for cells in layer.sets:
            n = len(cells)
            units, shared, filt = layer.unit[cells], layer.shared[cells], layer.filter[cells]
            b = api.Batch(pr.e, filt, shared)
            b.set_units(units)
            d_rng, d_pl = api.DeviceBuffer(pr.e, 8 * n), api.DeviceBuffer(pr.e, 16 * n)
            for kind, K in table[li]:
                it = 0
                while (L.les_random_is_continued(it, K, 0, mind, maxd) if kind == api.PROPOSE_RANDOM else it < K):
                    d_lab.upload(rl); d_cur.upload(rc); d_rng.upload(np.ascontiguousarray(st[cells]))
                    b.propose(kind, d_lab.ptr, d_rng.ptr, d_pl.ptr, m=it)
                    b.run(d_pl.ptr, d_prop.ptr, check=True, planes_on_device=True)
                    b.wta(d_pl.ptr, d_cur.ptr, d_prop.ptr, d_lab.ptr)
                    pr.e.synchronize()
                    planes, new_st = _oracle_proposals(kind, rl, W, units, st[cells], it, mind, maxd)
                    prop = pr.o.unary_batch(filt, shared, planes, check=True)
                    old_c = rc.copy()
                    for r, p in zip(shared, planes):
                        L.les_oracle_wta_update(W, om.Rect(*[int(v) for v in r]), rc.ctypes.data_as(C.c_void_p),
                                                prop.ctypes.data_as(C.c_void_p), rl.ctypes.data_as(C.c_void_p),
                                                om.Plane(*[float(v) for v in p]))
                    gp = d_pl.download((n,), api.PLANE_DT)
                    g4, r4 = gp.view(np.float32).reshape(n, 4), planes.view(np.float32).reshape(n, 4)
                    if plane_exact:
                        assert gp.tobytes() == planes.tobytes(), f"proposals differ (kind {kind})"
                        assert np.array_equal(d_rng.download((n,), np.uint64), new_st)
                    elif kind != api.PROPOSE_RANDOM:
                        assert gp.tobytes() == planes.tobytes(), f"proposals differ (kind {kind})"      # no trigonometry: exact on hardware too
                    else:
                        ok = np.all(np.abs(g4 - r4) <= 1e-4 * np.maximum(1, np.abs(r4)), axis=1)
                        assert ok.all(), f"proposal agreement {ok.mean()} (kind {kind})"
                    if gp.tobytes() == planes.tobytes():
                        gprop = d_prop.download((H, W), np.float32)
                        worst = max(worst, compare_maps(np.where(np.isnan(prop), np.nan, gprop).astype(np.float32), prop))
                        gl, gc = d_lab.download((H, W), api.PLANE_DT), d_cur.download((H, W), np.float32)
                        decisive = np.isnan(prop) | (np.abs(old_c - np.nan_to_num(prop, nan=0.0)) > 1e-5)
                        assert np.all((gl == rl)[decisive]), "WTA decision differs away from a tie"
                        assert np.max(np.abs(gc - rc)[rc < 1e5], initial=0.0) <= 1e-5
                    st[cells] = new_st
                    steps += 1
                    it += 1
            d_rng.free(); d_pl.free(); b.destroy()
    for d in (d_lab, d_cur, d_prop):
        d.free()

    # ---- (b) free-running device iteration vs oracle-driven iteration: same energy
    _, rc2 = pm_iteration_oracle(pr, layers_units, table, [s.copy() for s in seeds], labels0.copy(), cur0.copy(), 0)
    _, gc2 = pm_iteration_device(pr, layers_units, table, [s.copy() for s in seeds], labels0.copy(), cur0.copy(), 0)
    e_ref, e_got = float(rc2[rc2 < 1e5].sum()), float(gc2[gc2 < 1e5].sum())
    assert (rc2 < 1e5).mean() > 0.95 and (gc2 < 1e5).mean() > 0.95
    assert abs(e_got - e_ref) <= 0.02 * e_ref, (e_got, e_ref)
    return steps, worst


def case_volume_preparation(pr, lib):
    """N3: fillOutOfView / convertVolumeL2R on the device vs the oracle (bit exact), LES/main.cpp:146-199."""
    import ctypes as C
    L = om.lib()
    rng = np.random.default_rng(5)
    D, H, W = 9, 7, 40
    vol = rng.random((D, H, W), dtype=np.float32)
    src = api.DeviceBuffer(pr.e, vol.nbytes)
    dst = api.DeviceBuffer(pr.e, vol.nbytes)
    for mode in (0, 1):
        ref = vol.copy()
        L.les_fill_out_of_view(ref.ctypes.data_as(C.c_void_p), D, H, W, mode)
        src.upload(vol)
        api.fill_out_of_view(src.ptr, D, H, W, mode, lib=lib)
        pr.e.synchronize()
        assert src.download((D, H, W), np.float32).tobytes() == ref.tobytes()
    ref = np.zeros_like(vol)
    L.les_convert_volume_l2r(vol.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p), D, H, W)
    src.upload(vol)
    api.convert_volume_l2r(src.ptr, dst.ptr, D, H, W, lib=lib)
    pr.e.synchronize()
    assert dst.download((D, H, W), np.float32).tobytes() == ref.tobytes()
    src.free(); dst.free()


# ------------------------------------------------------------------------------------------------
# end-to-end quality on real data (scaled to what can travel to the GPU box)
# ------------------------------------------------------------------------------------------------
def cones_ad_volume(D=64):
    """Truncated absolute-difference matching cost between the cones crop and its (wider) right view:
    vol[d][y][x] = mean_c |imL(y,x,c) - imR(y,x-d,c)| / 255.  Returns (imL, vol, gt)."""
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "cones_crop.npz"))
    imL, imRw, gt = z["imL"], z["imR_wide"], z["gt"]
    H, W = imL.shape[:2]
    L = imL.astype(np.float32)
    vol = np.empty((D, H, W), np.float32)
    for d in range(D):
        R = imRw[:, 64 - d:64 - d + W].astype(np.float32)
        vol[d] = np.abs(L - R).mean(axis=2) / 255.0
    return imL, vol, gt


def case_quality_cones(lib, device, iters=3):
    """PatchMatch iterations (MiddV2 layer set-up, LES/main.cpp:300-306, without the graph cut) on a crop of the
    reference's bundled cones pair: the label map must converge towards the ground truth (Evaluator semantics,
    LES/Evaluator.h:133-140: bad pixel = |d - gt| > threshold where gt is known)."""
    import torch
    from localexpstereo_amd import pm
    imL, vol, gt = cones_ad_volume()
    e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=20, eps=1e-4, th_col=0.12, max_disp=63.0, lib=lib)
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)],
             [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
    r = pm.PMRunner(e, (5, 15, 25), table, seed=11, device=device)
    known = gt > 0

    def bad(thr):
        d = r.disparities().cpu().numpy()
        return float((np.abs(d - gt)[known] > thr).mean() * 100)

    r.init_labels()
    e.synchronize()
    hist = [(bad(1.0), float(r.cur.sum()))]
    for it in range(iters):
        r.iteration(it)
        e.synchronize()
        hist.append((bad(1.0), float(r.cur.sum())))
    r.close()
    e.close()
    assert hist[0][0] > 80.0                                     # random initial labels
    assert all(b[1] <= a[1] + 1e-3 for a, b in zip(hist, hist[1:])), hist      # WTA never increases the energy
    assert hist[-1][0] < 20.0, hist                              # converged to the surface almost everywhere
    return hist


def case_quality_cones_naive(lib, device, iters=2):
    """Config 1's own energy (NaiveStereoEnergy, MiddV2 parameters LES/main.cpp:86-121, layers :300-306) driven by the
    device-resident PatchMatch iterations on the cones crop.  The left crop is padded by 64 columns so that both
    views have the width of the stored wide right crop; bad pixels are counted on the original columns only."""
    from localexpstereo_amd import pm
    z = np.load(os.path.join(GOLDEN, "cones_crop.npz"))
    imL, imRw, gt = z["imL"], z["imR_wide"], z["gt"]
    H, W = imL.shape[:2]
    imLw = np.concatenate([np.repeat(imL[:, :1], 64, axis=1), imL], axis=1)
    e = api.HipCostVolumeEnergy.naive(imLw, np.ascontiguousarray(imRw), windR=20, eps=1e-4, alpha=0.9, th_col=10.0, th_grad=2.0,
                                      max_disp=63.0, lib=lib)
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)],
             [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
    r = pm.PMRunner(e, (5, 15, 25), table, seed=11, device=device)
    known = gt > 0

    def bad(thr):
        d = r.disparities().cpu().numpy()[:, 64:]
        return float((np.abs(d - gt)[known] > thr).mean() * 100)

    r.init_labels()
    e.synchronize()
    hist = [(bad(1.0), float(r.cur.sum()))]
    for it in range(iters):
        r.iteration(it)
        e.synchronize()
        hist.append((bad(1.0), float(r.cur.sum())))
    r.close()
    e.close()
    assert hist[0][0] > 80.0
    assert all(b[1] <= a[1] + 1e-3 for a, b in zip(hist, hist[1:])), hist
    assert hist[-1][0] < 20.0, hist
    return hist


def case_post_process(pr, seed=31, windR=None):
    """Dual-view post-processing (LES/PMStereoBase.h:111-256): consistency masks and the post-processed label maps must
    be bit-identical to the oracle.  Label maps: piecewise-planar scene seen from both views (so that most pixels are
    consistent), plus blocks of wrong labels, an occlusion band and image-border cases."""
    H, W = pr.H, pr.W
    windR = pr.e.params.windR if windR is None else windR
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)

    def scene(sign):
        lab = np.zeros((H, W, 4), np.float32)
        # three surfaces with disparities ~ 4, 9, 14; the right view sees them shifted by their disparity
        for k, (a, b, c) in enumerate([(0.01, 0.0, 4.0), (0.0, 0.02, 8.0), (-0.01, 0.01, 14.0)]):
            m = (xs + (0 if sign > 0 else c)) // (W / 3.0) == k if k < 2 else (xs + (0 if sign > 0 else c)) // (W / 3.0) >= 2
            # plane in this view's coordinates: d(x) for the right view is the left plane evaluated at x + d ~ x + c
            cc = c + (a * c if sign < 0 else 0.0)
            lab[m] = (a, b, cc, 0.0)
        return lab

    LL, LR = scene(+1.0), scene(-1.0)
    for lab in (LL, LR):
        for _ in range(6):                                     # blocks of outliers
            x0, y0 = int(rng.integers(0, W - 8)), int(rng.integers(0, H - 8))
            w, h = int(rng.integers(2, 14)), int(rng.integers(2, 10))
            lab[y0:y0 + h, x0:x0 + w] = (rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0, pr.D), 0.0)
        noisy = rng.random((H, W)) < 0.02                      # isolated outliers
        lab[noisy, 2] += rng.uniform(3, 9, int(noisy.sum())).astype(np.float32)
    LL[:, :3] = (0.0, 0.0, 1e12, 0.0)                          # huge disparity: maps far outside
    LL[5, 7] = (np.nan, 0.0, 1.0, 0.0)
    dl = LL[..., 0] * xs + LL[..., 1] * ys + LL[..., 2]
    dr = LR[..., 0] * xs + LR[..., 1] * ys + LR[..., 2]
    fl, fr = om.consistency_check(dl, dr, 1.5)
    bufs = [api.DeviceBuffer(pr.e, H * W * 16) for _ in range(2)] + [api.DeviceBuffer(pr.e, H * W) for _ in range(2)]
    bufs[0].upload(LL); bufs[1].upload(LR)
    pr.e.consistency_check(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, 1.5)
    gl, gr = bufs[2].download((H, W), np.uint8), bufs[3].download((H, W), np.uint8)
    assert np.array_equal(gl, fl) and np.array_equal(gr, fr)
    assert 0.02 < (fl > 0).mean() < 0.6 and (fl == 128).any() and (fl == 255).any()
    imL, imR = pr.e.imL, pr.e.imR
    for thr in (1.5, 1.0):
        ref = om.post_process(LL, LR, imL, imR, windR=windR, threshold=thr, omega=10.0)
        got = pr.e.post_process_host(LL, LR, threshold=thr, omega=10.0)
        for g, r, name in zip(got, ref, "LR"):
            same = (g.view(np.uint32) == r.view(np.uint32)).all(axis=2)
            assert same.all(), f"post-processed labels differ in view {name}: {int((~same).sum())} pixels (threshold {thr})"
        changed = (ref[0].view(np.uint32) != LL.view(np.uint32)).any(axis=2).mean()
        assert changed > 0.01
    for b in bufs:
        b.free()
    return float(changed)


def case_ingest_files(lib, device, tmp_path, D=12, H=20, W=37):
    """MiddV3 volume ingest (LES/main.cpp:353-368) from raw .acrt files: with and without im1.acrt."""
    from localexpstereo_amd import io as lio
    rng = np.random.default_rng(5)
    vl = rng.uniform(0, 1, (D, H, W)).astype(np.float32)
    vr = rng.uniform(0, 1, (D, H, W)).astype(np.float32)
    lio.save_cost_volume(str(tmp_path / "im0.acrt"), vl)
    for have_right in (True, False):
        if have_right:
            lio.save_cost_volume(str(tmp_path / "im1.acrt"), vr)
        elif os.path.exists(tmp_path / "im1.acrt"):
            os.remove(tmp_path / "im1.acrt")
        a = lio.load_cost_volume(str(tmp_path / "im0.acrt"), D, H, W)
        b = lio.load_cost_volume(str(tmp_path / "im1.acrt"), D, H, W)
        assert (b is not None) == have_right
        tl, tr = lio.ingest_volumes(a, b, device=device, lib=lib)
        rl = vl.copy()
        om.fill_out_of_view(rl, 0)
        rr = vr.copy() if have_right else om.convert_volume_l2r(rl)
        om.fill_out_of_view(rr, 1)
        assert np.array_equal(tl.cpu().numpy(), rl) and np.array_equal(tr.cpu().numpy(), rr)


def case_quality_cones_gc(lib, device, pm_iters=1, gc_iters=1, units=(5, 15, 25), lambda_=1.0, device_cuts=None, table=None, check_quality=True):
    """Local expansion moves proper on the cones crop: PatchMatch iteration(s), then graph-cut iterations whose
    proposals / unary costs come from the library under test and whose cuts run in liblocalexp_host.so.
    Checks: the reference's flow == energy self-check on every move (LES/FastGCStereo.h:561-594, <= 1e-5 relative),
    the total energy (data + smoothness) never increases, the error rate stays converged."""
    from localexpstereo_amd import gc as lgc
    from localexpstereo_amd import pm
    imL, vol, gt = cones_ad_volume()
    e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=20, eps=1e-4, th_col=0.12, max_disp=63.0, lib=lib)
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)],
             [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]][: len(units)] if table is None else table
    r = pm.PMRunner(e, units, table, seed=11, device=device)
    g = lgc.GraphCut(imL, None, lambda_=lambda_)
    known = gt > 0

    def bad(thr):
        d = r.disparities().cpu().numpy()
        return float((np.abs(d - gt)[known] > thr).mean() * 100)

    r.init_labels()
    for it in range(pm_iters):
        r.iteration(it)
    if device_cuts is not None:
        r.device_cuts = device_cuts
    r.begin_gc(g)
    hist = [(bad(1.0), g.data_cost(0), g.smoothness_cost(0))]
    for it in range(gc_iters):
        r.gc_iteration(it, check=(it == 0 and not device_cuts))          # later iterations: graph capacities computed on the device
        r.sync_gc_state()
        hist.append((bad(1.0), g.data_cost(0), g.smoothness_cost(0)))
        assert np.array_equal(r.labels.cpu().numpy(), g.labels[0])
    gap = r.gc_max_gap
    if device_cuts:
        assert r.gc_seconds.get("cells_cut_on_device", 0) > 0
    r.close(); e.close(); g.close()
    assert gap <= 1e-5, gap
    en = [h[1] + h[2] for h in hist]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(en, en[1:])), hist
    assert hist[-1][2] < hist[0][2], hist                                       # the smoothness term went down
    if check_quality:
        assert hist[-1][0] < 20.0, hist
    return hist, gap


def case_gc_sets_without_round_trips(lib, device, monkeypatch, units=(12,)):
    """pm.PMRunner._gc_set_without_round_trips (the finest layer: all proposals of a disjoint set enqueued without a host round trip, one failure word read
    per set) against the lock-step-by-lock-step path: bit-identical labels and costs.  And the roll-back: with the device solver's iteration limit forced to
    1 every cell fails, every set is rolled back and repeated on the slow path, whose host cuts must give the run with host cuts only."""
    from localexpstereo_amd import gc as lgc
    from localexpstereo_amd import pm
    imL, vol, gt = cones_ad_volume()
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 2)]]

    def run(mode):
        for k in ("LES_GC_PER_LOCKSTEP_CHECK", "LES_HIP_MAXFLOW_MAX_ITER"):
            monkeypatch.delenv(k, raising=False)
        if mode == "per_lockstep":
            monkeypatch.setenv("LES_GC_PER_LOCKSTEP_CHECK", "1")
        if mode == "rollback":
            monkeypatch.setenv("LES_HIP_MAXFLOW_MAX_ITER", "1")
        e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=20, eps=1e-4, th_col=0.12, max_disp=63.0, lib=lib)
        r = pm.PMRunner(e, units, table, seed=11, device=device)
        r.speculative_sets_on_cpu = True
        g = lgc.GraphCut(imL, None, lambda_=1.0)
        r.init_labels()
        r.iteration(0)
        r.device_cuts = "none" if mode == "host" else "all"
        r.begin_gc(g)
        r.gc_iteration(0)
        r.sync_gc_state()
        out = (r.labels.cpu().numpy().copy(), r.cur.cpu().numpy().copy(), dict(r.gc_seconds), [sh.rng.cpu().numpy().copy() for sh in r.shards[0]])
        r.close(); e.close(); g.close()
        return out

    lab_a, cur_a, sec_a, rng_a = run("per_lockstep")
    lab_b, cur_b, sec_b, rng_b = run("speculative")
    assert sec_a.get("sets_without_round_trips", 0) == 0 and sec_b.get("sets_without_round_trips", 0) > 0 and sec_b.get("sets_rolled_back", 0) == 0, (sec_a, sec_b)
    assert lab_a.tobytes() == lab_b.tobytes() and cur_a.tobytes() == cur_b.tobytes()
    assert all(np.array_equal(x, y) for x, y in zip(rng_a, rng_b))
    lab_c, cur_c, sec_c, rng_c = run("rollback")
    lab_d, cur_d, sec_d, rng_d = run("host")
    assert sec_c.get("sets_rolled_back", 0) > 0 and sec_c.get("sets_without_round_trips", 0) == 0, sec_c
    assert lab_c.tobytes() == lab_d.tobytes() and cur_c.tobytes() == cur_d.tobytes()
    assert all(np.array_equal(x, y) for x, y in zip(rng_c, rng_d))
    for k in ("LES_GC_PER_LOCKSTEP_CHECK", "LES_HIP_MAXFLOW_MAX_ITER"):
        monkeypatch.delenv(k, raising=False)
    return sec_b["sets_without_round_trips"], sec_c["sets_rolled_back"]


def case_device_cuts_vs_host_cuts(lib, device, gc_iters=2, units=(14, 43)):
    """(1) Lock-step by lock-step, on the graphs of real graph-cut iterations: the cells cut on the device
    (les_hip_batch_solve_graphs) against the host solver on the same payload.  Both are minimum cuts of the same float graphs
    with the same segment rule: the flow values agree to 1e-5, and the masks agree except for nodes on a tie that float
    rounding of the residuals may move (measured on the MI355X: 3 of 779 100 nodes in the hardest lock-steps; bound: 2e-5).
    (2) Whole iterations with device cuts against whole iterations with host cuts: the trajectories are both valid (a moved
    tie changes later proposals), so only the energies are compared (5e-3)."""
    from localexpstereo_amd import gc as lgc
    from localexpstereo_amd import pm
    imL, vol, gt = cones_ad_volume()
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]][: len(units)]
    energies, worst, checked = [], 0.0, 0
    for dev_cuts in (False, True):
        e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=20, eps=1e-4, th_col=0.12, max_disp=63.0, lib=lib)
        r = pm.PMRunner(e, units, table, seed=11, device=device)
        g = lgc.GraphCut(imL, None, lambda_=1.0)
        r.init_labels()
        r.iteration(0)
        r.device_cuts = dev_cuts
        r.begin_gc(g)
        if dev_cuts:
            # (1) on the state the iterations start from
            p = g.params
            for sh in r.shards[0][:6]:
                if not sh.n:
                    continue
                assert sh.batch.max_cell_nodes <= api.Batch.MAXFLOW_MAX_NODES
                r._gc_buffers(sh)
                for kind in (api.PROPOSE_EXPANSION, api.PROPOSE_RANDOM, api.PROPOSE_RANSAC):
                    sh.batch.propose(kind, r.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr(), m=0)
                    sh.batch.run(sh.planes.data_ptr(), r.prop.data_ptr(), mode=0, check=True, planes_on_device=True)
                    sh.batch.expansion_graph(sh.planes.data_ptr(), r.labels.data_ptr(), r.cur.data_ptr(), r.prop.data_ptr(), sh.payload.data_ptr(),
                                             mode=0, lambda_=p["lambda_"], th_smooth=p["th_smooth"], omega=p["omega"], epsilon=p["epsilon"])
                    st = api.DeviceBuffer(e, 4 * sh.n)
                    fl = api.DeviceBuffer(e, 8 * sh.n)
                    sh.batch.solve_graphs(sh.payload.data_ptr(), sh.masks.data_ptr(), st.ptr, fl.ptr)
                    e.synchronize()
                    assert not st.download((sh.n,), np.int32).any()
                    nn = sh.graph_nodes
                    dm = sh.masks[:nn].cpu().numpy()
                    ph = sh.payload[: nn * 5].cpu().numpy()
                    hm, hf = np.zeros(nn, np.uint8), np.zeros(sh.n, np.float64)
                    lgc.solve_prebuilt(sh.regions, ph, sh.graph_off, hm, flows_out=hf)
                    df = fl.download((sh.n,), np.float64)
                    # float residuals: every push rounds at the magnitude of the capacity it is taken from (1e6 terminals of
                    # invalid labels: ulp 0.06), so the two flow values agree relative to the total terminal capacity of the cell
                    tsum = np.add.reduceat(np.abs(ph.reshape(-1, 5)[:, 0]).astype(np.float64), np.asarray(sh.graph_off, np.int64))
                    tol = 1e-6 * tsum + 1e-5 * np.abs(hf) + 1e-5
                    assert (np.abs(df - hf) <= tol).all(), f"flow values differ by up to {np.abs(df - hf).max():.3e} (tolerance {tol[np.argmax(np.abs(df - hf))]:.3e})"
                    frac = float(((dm != 0) != (hm != 0)).mean())
                    worst = max(worst, frac)
                    checked += 1
                    assert frac <= 2e-5, f"{frac:.2e} of the nodes of a lock-step differ between the device cut and the host cut"
                    st.free(); fl.free()
        for it in range(gc_iters):
            r.gc_iteration(it)
        r.sync_gc_state()
        energies.append(g.data_cost(0) + g.smoothness_cost(0))
        assert (r.gc_seconds.get("cells_cut_on_device", 0) > 0) == dev_cuts
        r.close(); e.close(); g.close()
    assert checked >= 9
    assert abs(energies[0] - energies[1]) <= 5e-3 * abs(energies[0]), energies
    return worst, energies


def case_device_maxflow_edge_cells(pr, seed=3, tiled=False, kind=None):
    """les_hip_batch_solve_graphs on hand-made graphs of awkward shapes -- single rows and columns, 1 x 1 and 2 x 2 cells, cells with no
    arcs, with only source or only sink terminals, with terminals of 1e6 next to capacities below 1, and one cell just under the
    node limit -- against the host solver on the same payload: identical masks, equal flows.  Also: a cell above the limit is refused."""
    from localexpstereo_amd import gc as lgc
    rng = np.random.default_rng(seed)
    H, W = pr.H, pr.W
    shapes = [(1, 1), (2, 2), (1, 17), (19, 1), (7, 5), (12, 12), (16, 9), (5, 23), (13, 13), (11, 8), (9, 9), (10, 6)]
    if W >= 48 and H >= 48 and kind != 0:
        shapes.append((48, 48))                              # 2304 nodes: the largest cell les_maxflow.h takes (five nodes per thread)
    elif W >= 46 and H >= 68:
        shapes += [(46, 44), (30, 68), (45, 45)]            # the largest cells les_maxflow_cell.h takes: (w + 2) (h + 2) <= 2304, w h <= 2048, h <= 70
        if W >= 440:
            shapes += [(250, 7), (440, 3), (3, 70)]          # ... and its widest / tallest: the row of a node comes from one reciprocal (exactness argued in the header)
    rects, x, y, rowh = [], 0, 0, 0
    for (w, h) in shapes:
        if x + w > W:
            x, y, rowh = 0, y + rowh, 0
        if y + h > H:                          # (the solvers only look at the payload: cells may overlap in the image)
            x, y, rowh = 0, 0, 0
        assert w <= W and h <= H
        rects.append((x, y, w, h))
        x += w
        rowh = max(rowh, h)
    trs = api._rects(np.array(rects, np.int32))
    batch = api.Batch(pr.e, trs, trs)
    off, nn, k = batch.graph_offsets(), batch.graph_nodes(), len(rects)
    pay = np.zeros((nn, 5), np.float32)
    for i, (_, _, w, h) in enumerate(rects):
        n = w * h
        p = pay[off[i]: off[i] + n]
        p[:, 0] = rng.normal(0, 0.8, n)
        p[:, 1:] = rng.uniform(0, 0.6, (n, 4)) * (rng.uniform(0, 1, (n, 4)) < 0.8)
        variant = i % 6
        if variant == 1:
            p[:, 1:] = 0                                      # no arcs at all
        elif variant == 2:
            p[:, 0] = np.abs(p[:, 0])                         # only source terminals: everything takes the proposal
        elif variant == 3:
            p[:, 0] = -np.abs(p[:, 0])                        # only sink terminals: nothing changes
        elif variant == 4:
            big = rng.uniform(0, 1, n) < 0.3
            p[big, 0] = np.where(rng.uniform(0, 1, int(big.sum())) < 0.5, 1e6, -1e6)
        q = p.reshape(h, w, 5)                                # arcs E, S, SW, SE that would leave the cell carry no capacity
        q[:, -1, 1] = 0; q[-1, :, 2] = 0; q[-1, :, 3] = 0; q[:, 0, 3] = 0; q[-1, :, 4] = 0; q[:, -1, 4] = 0
    pay = np.ascontiguousarray(pay.reshape(-1))
    dp, dm, ds, df = api.DeviceBuffer(pr.e, nn * 20), api.DeviceBuffer(pr.e, nn), api.DeviceBuffer(pr.e, 4 * k), api.DeviceBuffer(pr.e, 8 * k)
    dp.upload(pay)
    if tiled:                                                 # the region-parallel solver (any cell size) on the same awkward shapes
        ws = api.DeviceBuffer(pr.e, batch.tiled_workspace_bytes())
        batch.solve_graphs_tiled(dp.ptr, dm.ptr, ds.ptr, ws.ptr, ws.nbytes, df.ptr)
        ws.free()
    else:
        assert kind is None or batch.graph_solver_kind == kind, (batch.graph_solver_kind, kind)
        batch.solve_graphs(dp.ptr, dm.ptr, ds.ptr, df.ptr)
    pr.e.synchronize()
    assert not ds.download((k,), np.int32).any()
    dev_m, dev_f = dm.download((nn,), np.uint8), df.download((k,), np.float64)
    host_m, host_f = np.zeros(nn, np.uint8), np.zeros(k, np.float64)
    lgc.solve_prebuilt(trs, pay, off, host_m, flows_out=host_f)
    assert np.array_equal(dev_m != 0, host_m != 0), f"{int(((dev_m != 0) != (host_m != 0)).sum())} nodes differ from the host cut"
    tsum = np.array([np.abs(pay.reshape(-1, 5)[off[i]: off[i] + w * h, 0]).astype(np.float64).sum() for i, (_, _, w, h) in enumerate(rects)])
    assert (np.abs(dev_f - host_f) <= 1e-6 * tsum + 1e-5 * np.abs(host_f) + 1e-5).all(), np.abs(dev_f - host_f).max()
    assert (dev_m != 0).any() and not (dev_m != 0).all()
    for b_ in (dp, dm, ds, df):
        b_.free()
    batch.destroy()
    if W >= 49 and H >= 48 and not tiled:                     # 49 x 48 > 2304 nodes: refused by the one-workgroup kernel (the tiled solver takes any size)
        big = api._rects(np.array([(0, 0, 49, 48)], np.int32))
        b2 = api.Batch(pr.e, big, big)
        assert b2.max_cell_nodes == 49 * 48
        d1, d2, d3 = api.DeviceBuffer(pr.e, 49 * 48 * 20), api.DeviceBuffer(pr.e, 49 * 48), api.DeviceBuffer(pr.e, 4)
        try:
            b2.solve_graphs(d1.ptr, d2.ptr, d3.ptr)
            raise AssertionError("a cell above LES_HIP_MAXFLOW_MAX_NODES was accepted")
        except api.LesHipError as ex:
            assert "exceeds the limit" in str(ex)
        for b_ in (d1, d2, d3):
            b_.free()
        b2.destroy()


def _grid_graph_reference(pay, w, h):
    """Independent min cut of one device-format cell (5 floats per node: terminal residual, caps E, S, SW, SE) with networkx:
    -> (max-flow value through the arcs, canonical SOURCE mask = nodes that can NOT reach the sink in the residual graph, which is the
    unique minimum cut with the smallest sink side: the rule of the reference's solver with SOURCE as the default, LES/FastGCStereo.h:557)."""
    import networkx as nx
    q = pay.reshape(h, w, 5).astype(np.float64)
    G = nx.DiGraph()
    G.add_nodes_from(["s", "t"])
    G.add_nodes_from(range(w * h))
    for y in range(h):
        for x in range(w):
            i = y * w + x
            tr = q[y, x, 0]
            if tr > 0:
                G.add_edge("s", i, capacity=tr)
            elif tr < 0:
                G.add_edge(i, "t", capacity=-tr)
            for k, (dx, dy) in enumerate(((1, 0), (0, 1), (-1, 1), (1, 1))):
                c = q[y, x, 1 + k]
                xx, yy = x + dx, y + dy
                if c > 0 and 0 <= xx < w and 0 <= yy < h:
                    G.add_edge(i, yy * w + xx, capacity=c)
    R = nx.algorithms.flow.preflow_push(G, "s", "t")
    value = R.graph["flow_value"]
    # nodes that reach t through arcs with residual capacity
    reach, stack = {"t"}, ["t"]
    while stack:
        v = stack.pop()
        for u in R.predecessors(v):
            if u not in reach and R[u][v]["capacity"] - R[u][v]["flow"] > 0:
                reach.add(u)
                stack.append(u)
    src = np.array([i not in reach for i in range(w * h)], bool)
    return value, src


def _cut_capacity(pay, w, h, src):
    """capacity of the s/t cut whose SOURCE side is `src` (bool per node), arcs + terminals, in double"""
    q = pay.reshape(h, w, 5).astype(np.float64)
    m = src.reshape(h, w)
    tr = q[..., 0]
    cap = np.where(m, np.maximum(-tr, 0), np.maximum(tr, 0)).sum()           # SOURCE nodes pay their sink link, SINK nodes their source link
    for k, (dx, dy) in enumerate(((1, 0), (0, 1), (-1, 1), (1, 1))):
        c = q[..., 1 + k]
        ys, xs = np.nonzero(c > 0)
        yy, xx = ys + dy, xs + dx
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        ys, xs, yy, xx = ys[ok], xs[ok], yy[ok], xx[ok]
        cap += (c[ys, xs] * (m[ys, xs] & ~m[yy, xx])).sum()
    return float(cap)


def _random_cell_payloads(rng, shapes, dyadic):
    """device-format payloads of random cells; dyadic: every capacity is a multiple of 2^-10 below 4, so that every flow and residual is
    exact in float32 and in double alike and the cut is unique up to genuine ties"""
    pays = []
    for (w, h) in shapes:
        n = w * h
        p = np.zeros((n, 5), np.float32)
        p[:, 0] = rng.normal(0, 0.8, n)
        p[:, 1:] = rng.uniform(0, 0.6, (n, 4)) * (rng.uniform(0, 1, (n, 4)) < 0.8)
        if dyadic:
            p = (np.round(p * 1024.0) / 1024.0).astype(np.float32)
        q = p.reshape(h, w, 5)
        q[:, -1, 1] = 0; q[-1, :, 2] = 0; q[-1, :, 3] = 0; q[:, 0, 3] = 0; q[-1, :, 4] = 0; q[:, -1, 4] = 0
        pays.append(p)
    return pays


def _solve_cells_on_device(pr, shapes, pays, tiled=False, poison=False, stats=None, kind=None):
    """tiled: the region-parallel solver for cells of any size (les_hip_batch_solve_graphs_tiled) instead of the one-workgroup-per-cell kernel.
    kind: which one-workgroup kernel the call must go to (0 = csrc/les_maxflow_cell.h, 1 / 2 = csrc/les_maxflow.h; None = not checked)."""
    H, W = pr.H, pr.W
    rects, x, y, rowh = [], 0, 0, 0
    for (w, h) in shapes:
        if x + w > W:
            x, y, rowh = 0, y + rowh, 0
        if y + h > H:                          # (the solvers only look at the payload: cells may overlap in the image)
            x, y, rowh = 0, 0, 0
        assert w <= W and h <= H, "a cell does not fit the image"
        rects.append((x, y, w, h))
        x += w
        rowh = max(rowh, h)
    trs = api._rects(np.array(rects, np.int32))
    batch = api.Batch(pr.e, trs, trs)
    off, nn, k = batch.graph_offsets(), batch.graph_nodes(), len(rects)
    pay = np.zeros((nn, 5), np.float32)
    for i, p in enumerate(pays):
        pay[off[i]: off[i] + len(p)] = p
    pay = np.ascontiguousarray(pay.reshape(-1))
    dp, dm, ds, df = api.DeviceBuffer(pr.e, nn * 20), api.DeviceBuffer(pr.e, nn), api.DeviceBuffer(pr.e, 4 * k), api.DeviceBuffer(pr.e, 8 * k)
    dp.upload(pay)
    if tiled:
        ws = api.DeviceBuffer(pr.e, batch.tiled_workspace_bytes())
        if poison:
            ws.fill(0xA5); dm.fill(0x5A); ds.fill(0x7F)
        batch.solve_graphs_tiled(dp.ptr, dm.ptr, ds.ptr, ws.ptr, ws.nbytes, df.ptr)
        if stats is not None:
            stats.update(batch.tiled_stats)
        ws.free()
    else:
        assert kind is None or batch.graph_solver_kind == kind, (batch.graph_solver_kind, kind)
        batch.solve_graphs(dp.ptr, dm.ptr, ds.ptr, df.ptr)
    pr.e.synchronize()
    status = ds.download((k,), np.int32)
    masks, flows = dm.download((nn,), np.uint8), df.download((k,), np.float64)
    for b_ in (dp, dm, ds, df):
        b_.free()
    batch.destroy()
    return off, status, masks, flows


def case_device_maxflow_vs_networkx(pr, seed=5, ncells=50, max_side=45, tiled=False, kind=None):
    """les_maxflow_kernel against an INDEPENDENT checker (networkx preflow-push + residual reachability), not against the host solver:
      * dyadic capacities (arithmetic exact in float and double): the device mask equals the canonical minimum cut node for node and the
        flow is equal;
      * arbitrary float32 capacities: the flow agrees to 1e-6 relative, the device mask is a minimum cut (its capacity equals the
        max-flow value to 1e-6), and every node where it differs from the canonical cut is a tie (moving those nodes changes the cut
        capacity by < 1e-6 of it).
    -> (cells, nodes, differing nodes in the float cases)"""
    rng = np.random.default_rng(seed)
    side_hi = min(max_side, pr.W, pr.H)
    total_nodes = total_diff = 0
    for dyadic in (True, False):
        shapes = [(int(rng.integers(6, side_hi + 1)), int(rng.integers(6, side_hi + 1))) for _ in range(ncells // 2)]
        shapes[0] = (side_hi, side_hi)
        if side_hi >= 42:
            shapes[1] = (42, 42)
        pays = _random_cell_payloads(rng, shapes, dyadic)
        off, status, masks, flows = _solve_cells_on_device(pr, shapes, pays, tiled=tiled, kind=kind)
        assert not status.any(), "a cell hit the iteration limit"
        for i, ((w, h), p) in enumerate(zip(shapes, pays)):
            ref_flow, ref_src = _grid_graph_reference(p, w, h)
            dev_src = masks[off[i]: off[i] + w * h] != 0
            scale = max(1.0, np.abs(p[:, 0]).astype(np.float64).sum())
            total_nodes += w * h
            if dyadic:
                assert np.array_equal(dev_src, ref_src), f"cell {i} ({w}x{h}): {int((dev_src != ref_src).sum())} nodes differ from the canonical minimum cut"
                assert abs(flows[i] - ref_flow) <= 1e-9 * scale, (flows[i], ref_flow)
            else:
                assert abs(flows[i] - ref_flow) <= 1e-6 * scale, (flows[i], ref_flow)
                cap_dev = _cut_capacity(p, w, h, dev_src)
                assert abs(cap_dev - ref_flow) <= 1e-6 * scale, f"cell {i}: the device mask is not a minimum cut ({cap_dev} vs {ref_flow})"
                total_diff += int((dev_src != ref_src).sum())
    return ncells // 2 * 2, total_nodes, total_diff


def case_device_maxflow_vs_brute_force(pr, seed=9, ncells=40, tiled=False, kind=None):
    """Cells of at most 4 x 4 nodes: every one of the 2^n labelings is enumerated; the device mask must be THE canonical minimum cut
    (the minimum-capacity labeling whose sink side is the intersection of all minimum sink sides).  Dyadic capacities: exact."""
    rng = np.random.default_rng(seed)
    shapes = [(int(rng.integers(1, 5)), int(rng.integers(1, 5))) for _ in range(ncells)]
    shapes[0] = (4, 4)
    pays = _random_cell_payloads(rng, shapes, dyadic=True)
    off, status, masks, flows = _solve_cells_on_device(pr, shapes, pays, tiled=tiled, kind=kind)
    assert not status.any()
    for i, ((w, h), p) in enumerate(zip(shapes, pays)):
        n = w * h
        codes = np.arange(1 << n, dtype=np.int64)
        bits = ((codes[:, None] >> np.arange(n)[None, :]) & 1).astype(bool)              # True = SOURCE
        q = p.reshape(h, w, 5).astype(np.float64)
        tr = q[..., 0].reshape(-1)
        cap = np.where(bits, np.maximum(-tr, 0)[None, :], np.maximum(tr, 0)[None, :]).sum(1)
        for k, (dx, dy) in enumerate(((1, 0), (0, 1), (-1, 1), (1, 1))):
            for y in range(h):
                for x in range(w):
                    c, xx, yy = q[y, x, 1 + k], x + dx, y + dy
                    if c > 0 and 0 <= xx < w and 0 <= yy < h:
                        cap += c * (bits[:, y * w + x] & ~bits[:, yy * w + xx])
        best = cap.min()
        minimal = bits[cap == best]
        # minimum cuts are closed under union of their source sides: the canonical cut (smallest sink side) is that union
        canonical = minimal.any(axis=0)
        assert cap[(bits == canonical[None, :]).all(1)][0] == best
        dev_src = masks[off[i]: off[i] + n] != 0
        assert np.array_equal(dev_src, canonical), f"cell {i} ({w}x{h}): device {dev_src.astype(int)} canonical {canonical.astype(int)}"
        assert abs(flows[i] - best) <= 1e-9 * max(1.0, np.abs(tr).sum()), (flows[i], best)
    return len(shapes)


def case_tiled_maxflow_large_cells(pr, seed=11, shapes=None):
    """les_hip_batch_solve_graphs_tiled on cells of SEVERAL tiles -- wide, tall, one-row, one-column, just over one tile, many tiles --
    against networkx (dyadic capacities: the canonical cut node for node, equal flow; float capacities: a minimum cut, flow to 1e-6) and
    against the host solver (identical masks on the dyadic cells).  The workspace is poisoned first: nothing may depend on its contents,
    and a second solve on the same workspace must return the same bytes (bit-reproducible).  -> (cells, nodes, tie nodes)"""
    from localexpstereo_amd import gc as lgc
    rng = np.random.default_rng(seed)
    if shapes is None:
        shapes = [(150, 130), (65, 31), (31, 65), (pr.W, 1), (1, min(pr.H, 300)), (129, 129), (64, 30), (200, 45)]
    shapes = [(min(w, pr.W), min(h, pr.H)) for (w, h) in shapes]
    total_nodes = ties = 0
    for dyadic in (True, False):
        pays = _random_cell_payloads(rng, shapes, dyadic)
        off, status, masks, flows = _solve_cells_on_device(pr, shapes, pays, tiled=True, poison=True)
        off2, status2, masks2, flows2 = _solve_cells_on_device(pr, shapes, pays, tiled=True)
        assert not status.any() and not status2.any()
        assert np.array_equal(masks, masks2), "two solves of the same lock-step differ"
        for i, ((w, h), p) in enumerate(zip(shapes, pays)):
            ref_flow, ref_src = _grid_graph_reference(p, w, h)
            dev_src = masks[off[i]: off[i] + w * h] != 0
            scale = max(1.0, np.abs(p[:, 0]).astype(np.float64).sum())
            total_nodes += w * h
            if dyadic:
                assert np.array_equal(dev_src, ref_src), f"cell {i} ({w}x{h}): {int((dev_src != ref_src).sum())} nodes differ from the canonical minimum cut"
                assert abs(flows[i] - ref_flow) <= 1e-9 * scale, (flows[i], ref_flow)
                hm = np.zeros(w * h, np.uint8)
                lgc.solve_prebuilt(api._rects(np.array([(0, 0, w, h)], np.int32)), np.ascontiguousarray(p.reshape(-1), np.float32), np.array([0], np.int64), hm, nthreads=1)
                assert np.array_equal(dev_src, hm != 0), f"cell {i}: differs from the host solver"
            else:
                assert abs(flows[i] - ref_flow) <= 1e-6 * scale, (flows[i], ref_flow)
                cap_dev = _cut_capacity(p, w, h, dev_src)
                assert abs(cap_dev - ref_flow) <= 1e-6 * scale, f"cell {i}: the device mask is not a minimum cut ({cap_dev} vs {ref_flow})"
                ties += int((dev_src != ref_src).sum())
    return 2 * len(shapes), total_nodes, ties


def case_tiled_maxflow_hard_cells(pr):
    """The two committed 129 x 129 crops of real coarse-layer lock-steps (tests/golden/hard_cells.npz: 89 % and 20 % of the nodes switch) through
    the tiled device solver: masks node for node equal to the host solver's (liblocalexp_host.so), flows equal to 1e-6.  -> nodes that switch"""
    import os
    from localexpstereo_amd import gc as lgc
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hard_cells.npz"))
    shapes, pays = [], []
    for k in z.files:
        h, w = z[k].shape[:2]
        shapes.append((w, h))
        pays.append(np.ascontiguousarray(z[k].reshape(-1, 5), np.float32))
    off, status, masks, flows = _solve_cells_on_device(pr, shapes, pays, tiled=True)
    assert not status.any()
    switched = 0
    for i, ((w, h), p) in enumerate(zip(shapes, pays)):
        hm, hf = np.zeros(w * h, np.uint8), np.zeros(1)
        lgc.solve_prebuilt(api._rects(np.array([(0, 0, w, h)], np.int32)), np.ascontiguousarray(p.reshape(-1)), np.array([0], np.int64), hm, nthreads=1, flows_out=hf)
        dev = masks[off[i]: off[i] + w * h] != 0
        assert np.array_equal(dev, hm != 0), f"cell {i}: {int((dev != (hm != 0)).sum())} nodes differ from the host cut"
        # (terminals of 1e6 -- invalid labels -- sit next to capacities below 1: a float excess of 0.01 absorbed by such a sink arc is below its ulp;
        # the bound is the one of the one-workgroup kernel's test, relative to the terminal capacities)
        tsum = float(np.abs(p[:, 0]).astype(np.float64).sum())
        assert abs(flows[i] - hf[0]) <= 1e-6 * tsum + 1e-5 * abs(hf[0]) + 1e-5, (flows[i], hf[0], tsum)
        assert 0 < dev.mean() < 1
        switched += int(dev.sum())
    return switched


def case_tiled_maxflow_handover(pr, monkeypatch, seed=13, shapes=None):
    """The hand-over of the tiled solver (csrc/les_maxflow_tiled.h, host/ResidualCut.h): with the threshold lowered so that cells ARE still open
    when the host looks (after the first 12 launches), the open cells' residual graphs go to the host cores, which finish them.  The masks must be
    those of the solve without hand-over and of the host solver on the original payload (dyadic capacities: node for node; float capacities: a
    minimum cut of the same value), the flow values must agree, for both host finishers (search trees / push-relabel).  Also the two committed
    hard crops.  -> cells handed over"""
    import os
    from localexpstereo_amd import gc as lgc
    rng = np.random.default_rng(seed)
    if shapes is None:
        shapes = [(150, 130), (65, 31), (129, 129), (200, 45), (31, 65), (64, 30)]
    shapes = [(min(w, pr.W), min(h, pr.H)) for (w, h) in shapes]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hard_cells.npz"))
    hard_shapes = [(z[k].shape[1], z[k].shape[0]) for k in z.files]
    hard_pays = [np.ascontiguousarray(z[k].reshape(-1, 5), np.float32) for k in z.files]
    handed = 0
    for name, shp, pays, dyadic in (("dyadic", shapes, _random_cell_payloads(rng, shapes, True), True), ("float", shapes, _random_cell_payloads(rng, shapes, False), False),
                                    ("hard", hard_shapes, hard_pays, False)):
        monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER", "0")
        off, status0, masks0, flows0 = _solve_cells_on_device(pr, shp, pays, tiled=True)
        assert not status0.any()
        for solver in (0, 1):
            monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER", "1")
            monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER_AFTER", "1")
            monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER_NODES", "1000000")
            monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER_NO_STALL_RULE", "1")   # (the product waits for a group of launches in which no cell finished)
            monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER_SOLVER", str(solver))
            monkeypatch.setenv("LES_GC_RESIDUAL_BAND_NODES", "2000")      # row bands inside the handed-over cells (a team inside the cell-per-thread team) at test sizes too
            st = {}
            off1, status1, masks1, flows1 = _solve_cells_on_device(pr, shp, pays, tiled=True, poison=True, stats=st)
            assert not status1.any() and np.array_equal(off, off1)
            assert st["handed_cells"] > 0 and st["handed_nodes"] > 0, f"{name}: nothing was handed over ({st})"
            handed += st["handed_cells"]
            for i, ((w, h), p) in enumerate(zip(shp, pays)):
                a, b = masks0[off[i]: off[i] + w * h] != 0, masks1[off[i]: off[i] + w * h] != 0
                tsum = max(1.0, float(np.abs(p[:, 0]).astype(np.float64).sum()))
                if dyadic:
                    assert np.array_equal(a, b), f"{name} cell {i} ({w}x{h}), solver {solver}: {int((a != b).sum())} nodes differ from the cut without hand-over"
                    assert abs(flows1[i] - flows0[i]) <= 1e-9 * tsum, (flows1[i], flows0[i])
                else:
                    assert abs(flows1[i] - flows0[i]) <= 1e-6 * tsum + 1e-5 * abs(flows0[i]) + 1e-5, (name, i, flows1[i], flows0[i])
                    if name == "hard":
                        hm = np.zeros(w * h, np.uint8)
                        lgc.solve_prebuilt(api._rects(np.array([(0, 0, w, h)], np.int32)), np.ascontiguousarray(p.reshape(-1)), np.array([0], np.int64), hm, nthreads=1)
                        assert np.array_equal(b, hm != 0), f"hard cell {i}, solver {solver}: {int((b != (hm != 0)).sum())} nodes differ from the host cut"
                    else:
                        ca, cb = _cut_capacity(p, w, h, a), _cut_capacity(p, w, h, b)
                        assert abs(ca - cb) <= 1e-6 * tsum, f"{name} cell {i}: the cut after hand-over is not a minimum cut ({cb} vs {ca})"
        for k in ("LES_HIP_MAXFLOW_HANDOVER", "LES_HIP_MAXFLOW_HANDOVER_AFTER", "LES_HIP_MAXFLOW_HANDOVER_SOLVER", "LES_GC_RESIDUAL_BAND_NODES", "LES_HIP_MAXFLOW_HANDOVER_NO_STALL_RULE", "LES_HIP_MAXFLOW_HANDOVER_NODES"):
            monkeypatch.delenv(k, raising=False)
    return handed


def case_refresh_volume(lib, H=90, W=130, D=10):
    """les_hip_refresh_volume: a context created on a DEVICE-resident volume keeps things derived from it (the cost range that fixes the march kernel's fixed-point
    scales, the tiled copy steep planes gather from).  After the caller refills the volume in place -- here with costs of another range, [-1, 2) instead of [0, 1) -- and
    calls refresh, the context must produce exactly what a context created on the new volume produces (bit for bit; fronto-parallel, slanted and steep planes)."""
    imL = synth.make_guide(H, W, 1234)
    volA = synth.make_volume(D, H, W, 42)
    volB = (synth.make_volume(D, H, W, 43) * np.float32(3.0) - np.float32(1.0)).astype(np.float32)
    planes = np.concatenate([synth.fronto_planes(D)[:4], synth.slanted_planes(6, H, W, D - 1, seed=7), random_planes(4, D, H, W, 3, slant=0.04)]).astype(np.float32)
    n = len(planes)
    full = api._rects(np.array([(0, 0, W, H)] * n, np.int32))

    def run(e):
        b = api.Batch(e, full, full, out_slabs=True)
        assert b.kernel_kind(0) == 1
        out = api.DeviceBuffer(e, n * H * W * 4)
        b.run(planes, out.ptr, mode=0, check=True)
        e.synchronize()
        got = out.download((n, H, W), np.float32)
        out.free(); b.destroy()
        return got
    helper = api.HipCostVolumeEnergy(imL, None, volA, None, windR=20, eps=1e-4, th_col=0.5, lib=lib)       # (owns nothing of interest: device allocations go through a context)
    dv = api.DeviceBuffer(helper, D * H * W * 4)
    dv.upload(volA)
    e = api.HipCostVolumeEnergy(imL, None, dv.ptr, None, windR=20, eps=1e-4, th_col=0.5, volumes_on_device=True, shape=(D, H, W), lib=lib)
    a_dev = run(e)
    a_ref = run(helper)
    assert np.array_equal(a_dev.view(np.uint32), a_ref.view(np.uint32))
    dv.upload(volB)                                            # the caller refills its volume in place ...
    e.refresh_volume(0)                                        # ... and says so
    b_dev = run(e)
    fresh = api.HipCostVolumeEnergy(imL, None, volB, None, windR=20, eps=1e-4, th_col=0.5, lib=lib)
    b_ref = run(fresh)
    assert np.array_equal(b_dev.view(np.uint32), b_ref.view(np.uint32)), f"{int((b_dev.view(np.uint32) != b_ref.view(np.uint32)).sum())} values differ after the refresh"
    assert not np.array_equal(a_dev, b_dev)
    # A context created on a PLACEHOLDER volume (NaN: the march kernel's preconditions fail, the strip kernel serves it) must reach the march kernel once the
    # caller has filled the volume and refreshed -- the guide's tables are built at creation whatever the volume holds (round 5 left such a context on
    # the 2.2 x slower strip kernel for good) -- and a refresh onto an unusable volume must fall back again and release the tiled copy.
    dv2 = api.DeviceBuffer(helper, D * H * W * 4)
    dv2.upload(np.full((D, H, W), np.nan, np.float32))
    e2 = api.HipCostVolumeEnergy(imL, None, dv2.ptr, None, windR=20, eps=1e-4, th_col=0.5, volumes_on_device=True, shape=(D, H, W), lib=lib)
    b0 = api.Batch(e2, full, full, out_slabs=True)
    assert b0.kernel_kind(0) == 0 and e2.tiled_volume_bytes(0) == 0
    b0.destroy()
    dv2.upload(volB)
    e2.refresh_volume(0)
    c_dev = run(e2)                                            # (asserts the march kernel)
    assert np.array_equal(c_dev.view(np.uint32), b_ref.view(np.uint32))
    dv2.upload(np.full((D, H, W), np.nan, np.float32))
    e2.refresh_volume(0)
    b0 = api.Batch(e2, full, full, out_slabs=True)
    assert b0.kernel_kind(0) == 0 and e2.tiled_volume_bytes(0) == 0, "a refresh onto an unusable volume kept the march kernel or the stale tiled copy"
    b0.destroy()
    e2.close(); dv2.free()
    fresh.close(); e.close(); dv.free(); helper.close()
    return n


def case_relative_error_floor(lib, th_col, H=240, W=320, D=32):
    """north_star states the tolerance as 1e-4 RELATIVE; the march kernel's error is ABSOLUTE (fixed-point steps that scale with th_col - vmin), so the
    relative claim has a floor: the cost below which 1e-4 relative is not met.  An absolute-difference style volume -- min(1, 0.12 |d - gt|) * 2 th_col, exact zeros where
    the ground truth is an integer, long near-zero ramps around it, no noise -- evaluated on every fronto-parallel plane and on the ground-truth planes themselves (whole
    regions aggregate to exactly 0 or to 1e-4 ... 1e-2 of th_col).  -> (max abs err, floor = largest oracle cost whose relative error exceeds 1e-4, number of evaluations
    below 1 % of th_col, their max abs err).  Asserted by the callers: abs err within the documented bound, floor <= 6e-3 th_col, sentinel / written-pixel sets exact."""
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    gt = np.where(xs < W // 2, 9.0, np.float32(0.03) * xs + np.float32(0.02) * ys + np.float32(4.0)).astype(np.float32)      # an integer half and a slanted half
    vol = np.empty((D, H, W), np.float32)
    for d in range(D):
        vol[d] = np.minimum(np.float32(1.0), np.float32(0.12) * np.abs(np.float32(d) - gt)) * np.float32(2.0 * th_col)
    assert (vol == 0).sum() > 1000
    imL, imR = synth.make_guide(H, W, 1234), synth.make_guide(H, W, 1235)
    e = api.HipCostVolumeEnergy(imL, imR, vol, vol, windR=20, eps=1e-4, th_col=th_col, lib=lib)
    o = om.Oracle(imL, imR, vol, vol, windR=20, eps=1e-4, th_col=th_col)
    planes = np.concatenate([synth.fronto_planes(D)[: D - 1], np.array([[0, 0, 9.0, 0], [0.03, 0.02, 4.0, 0], [0.03, 0.02, 4.25, 0], [0, 0, 9.5, 0]], np.float32)])
    n = len(planes)
    full = api._rects(np.array([(0, 0, W, H)] * n, np.int32))
    worst_abs, floor, nsmall, small_abs = 0.0, 0.0, 0, 0.0
    b = api.Batch(e, full, full, out_slabs=True)
    assert b.kernel_kind(0) == 1, "the march kernel must serve this batch"
    dout = api.DeviceBuffer(e, n * H * W * 4)
    b.run(planes, dout.ptr, mode=0, check=True)
    e.synchronize()
    got = dout.download((n, H, W), np.float32)
    ref = o.aggregate_planes(planes, mode=0, check=True)
    assert np.array_equal(got == 1e6, ref == 1e6), "invalid-label sentinels differ"
    v = ref != 1e6
    d = np.abs(got[v].astype(np.float64) - ref[v])
    r = ref[v].astype(np.float64)
    worst_abs = float(d.max())
    bad = d > 1e-4 * np.abs(r)
    floor = float(r[bad].max()) if bad.any() else 0.0
    sm = r < 0.01 * th_col
    nsmall, small_abs = int(sm.sum()), float(d[sm].max()) if sm.any() else 0.0
    zero = r == 0.0
    nzero, zero_abs = int(zero.sum()), float(d[zero].max()) if zero.any() else 0.0
    dout.free(); b.destroy(); e.close(); o.close() if hasattr(o, "close") else None
    return dict(max_abs_err=worst_abs, relative_floor=floor, evals_below_1pct_of_th=nsmall, max_abs_err_below_1pct=small_abs, exact_zero_costs=nzero, max_abs_err_on_exact_zeros=zero_abs)


def case_plain_build_equals_product(plain_lib, device="cuda", H=300, W=420, D=24):
    """The product library against libles_plain.so -- the same sources compiled with -DLES_MARCH_SCAN_PLAIN -DLES_MARCH_STATS_PLAIN -DLES_SIMT_PLAIN, i.e. with
    every inline-assembly path of the march kernel (the v_add_u32_dpp scan, the tied-destination buffer loads of role C with hand-kept vmcnt, the SDWA /
    cvt_rpi / med3 / mad_i64 primitives) replaced by plain C++: on whole-image slabs (fronto-parallel, slanted, steep = tiled-copy taps, NaN / huge planes),
    on LayerManager cell batches of two layers, on both views and on the image-based energy the outputs must be BIT-identical.  -> arrays compared"""
    import torch
    rng = np.random.default_rng(4)
    imL, imR = synth.make_guide(H, W, 1234), synth.make_guide(H, W, 1235)
    volL, volR = synth.make_volume(D, H, W, 42), synth.make_volume(D, H, W, 43)
    planes = np.concatenate([synth.fronto_planes(D)[:8], synth.slanted_planes(10, H, W, D - 1, seed=7), random_planes(6, D, H, W, 3, slant=0.04)]).astype(np.float32)
    planes[9] = (np.nan, 0.1, 3.0, 0.0); planes[10] = (1e30, 0.0, 0.0, 0.0); planes[11, 2] += 0.5
    n = len(planes)
    full = [(0, 0, W, H)] * n
    outs = {}
    for tag, lib in (("product", None), ("plain", plain_lib)):
        res = []
        e = api.HipCostVolumeEnergy(imL, imR, volL, volR, windR=20, eps=1e-4, th_col=0.5, lib=lib)
        out = torch.zeros((n, H, W), dtype=torch.float32, device=device)
        for mode in (0, 1):
            for check in (False, True):
                b = api.Batch(e, full, full, out_slabs=True)
                assert b.kernel_kind(mode) == 1, "the march kernel must serve this batch"
                out.fill_(-1.0)
                b.run(planes, out.data_ptr(), mode=mode, check=check)
                e.synchronize()
                res.append(out.cpu().numpy().copy())
                b.destroy()
        for unit in (14, 43):
            layer = om.Layer(W, H, 20, unit)
            for si in (0, len(layer.sets) // 2):
                cells = layer.sets[si]
                pl = random_planes(len(cells), D, H, W, 100 + si, slant=0.2)
                res.append(e.unary_batch(layer.filter[cells], layer.shared[cells], pl, mode=0, check=True).copy())
        e.close()
        en = api.HipCostVolumeEnergy.naive(imL, imR, windR=20, eps=1e-4, max_disp=float(D - 1), lib=lib)
        layer = om.Layer(W, H, 20, 25)
        cells = layer.sets[1]
        res.append(en.unary_batch(layer.filter[cells], layer.shared[cells], random_planes(len(cells), D, H, W, 9, slant=0.1), mode=0, check=True).copy())
        en.close()
        outs[tag] = res
    for i, (a, b) in enumerate(zip(outs["product"], outs["plain"])):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"output {i}: {int((a.view(np.uint32) != b.view(np.uint32)).sum())} values differ between the assembly and the plain build"
    return len(outs["product"])


def case_exchange_pack_unpack(pr, seed=2):
    """les_hip_exchange_pack / _unpack (the multi-GPU tile exchange behind the C ABI) against a numpy restatement of the slot layout:
    three ranks' rect lists, this process plays rank 1; its slot must hold its own tiles, and unpacking a gathered buffer built with
    numpy from ANOTHER pair of maps must overwrite exactly the other ranks' tiles and nothing else."""
    rng = np.random.default_rng(seed)
    H, W = pr.H, pr.W
    e = pr.e
    def rand_rects(n):
        out = []
        for _ in range(n):
            w, h = int(rng.integers(1, min(40, W))), int(rng.integers(1, min(30, H)))
            out.append((int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h))
        return np.array(out, np.int32).reshape(-1, 4)
    # disjoint tiles per rank are what the optimiser produces; overlapping ones are fine for pack and make unpack order-dependent, so
    # the test uses a tiling: vertical bands of the image, dealt to the ranks
    bands = np.linspace(0, W, 8).astype(int)
    rects = [[], [], []]
    for i in range(7):
        x0, x1 = int(bands[i]), int(bands[i + 1])
        ys = np.sort(rng.choice(np.arange(1, H), 2, replace=False))
        for (y0, y1) in ((0, int(ys[0])), (int(ys[0]), int(ys[1])), (int(ys[1]), H)):
            rects[int(rng.integers(0, 3))].append((x0, y0, x1 - x0, y1 - y0))
    rects[2].append((0, 0, 0, 0))                                             # an empty rect is legal
    rects = [np.array(r, np.int32).reshape(-1, 4) for r in rects]
    x = api.Exchange(e, 1, rects)
    npx = [int(sum(int(r[2]) * int(r[3]) for r in rr)) for rr in rects]
    lmax = (max(npx) + 3) // 4 * 4
    assert x.slot_floats == 5 * lmax
    lab = rng.normal(size=(H, W, 4)).astype(np.float32)
    cost = rng.normal(size=(H, W)).astype(np.float32)
    other_lab = rng.normal(size=(H, W, 4)).astype(np.float32)
    other_cost = rng.normal(size=(H, W)).astype(np.float32)

    def np_slot(rr, L, Cm):
        sl = np.zeros(5 * lmax, np.float32)
        off = 0
        for (rx, ry, rw, rh) in rr:
            n = int(rw) * int(rh)
            sl[4 * off: 4 * (off + n)] = L[ry:ry + rh, rx:rx + rw].reshape(-1)
            sl[4 * lmax + off: 4 * lmax + off + n] = Cm[ry:ry + rh, rx:rx + rw].reshape(-1)
            off += n
        return sl
    d_lab, d_cost = api.DeviceBuffer(e, lab.nbytes), api.DeviceBuffer(e, cost.nbytes)
    d_slot, d_recv = api.DeviceBuffer(e, 20 * lmax), api.DeviceBuffer(e, 60 * lmax)
    d_lab.upload(lab); d_cost.upload(cost); d_slot.fill(0)
    x.pack(d_lab.ptr, d_cost.ptr, d_slot.ptr)
    e.synchronize()
    got = d_slot.download((5 * lmax,), np.float32)
    ref = np_slot(rects[1], lab, cost)
    used = npx[1]
    assert np.array_equal(got[: 4 * used], ref[: 4 * used]) and np.array_equal(got[4 * lmax: 4 * lmax + used], ref[4 * lmax: 4 * lmax + used])
    gathered = np.concatenate([np_slot(rects[0], other_lab, other_cost), np.full(5 * lmax, np.nan, np.float32), np_slot(rects[2], other_lab, other_cost)])
    d_recv.upload(gathered)
    x.unpack(d_recv.ptr, d_lab.ptr, d_cost.ptr)
    e.synchronize()
    lab2, cost2 = d_lab.download((H, W, 4), np.float32), d_cost.download((H, W), np.float32)
    exp_lab, exp_cost = lab.copy(), cost.copy()
    for r in (0, 2):
        for (rx, ry, rw, rh) in rects[r]:
            exp_lab[ry:ry + rh, rx:rx + rw] = other_lab[ry:ry + rh, rx:rx + rw]
            exp_cost[ry:ry + rh, rx:rx + rw] = other_cost[ry:ry + rh, rx:rx + rw]
    assert np.array_equal(lab2, exp_lab) and np.array_equal(cost2, exp_cost)
    # errors: a rect outside the image, a first[] table that does not cover the rects
    try:
        api.Exchange(e, 0, [np.array([(W - 2, 0, 5, 5)], np.int32)])
        raise AssertionError("a rect outside the image was accepted")
    except api.LesHipError as ex:
        assert "outside the image" in str(ex)
    for b_ in (d_lab, d_cost, d_slot, d_recv):
        b_.free()
    x.destroy()
    return sum(npx)


def case_stereo_driver(lib, device, units=(16,), pmInit=1, maxIteration=1):
    """The Python FastGCStereo mirror end to end on the (padded) cones crop with config 1's energy, two views:
    PatchMatch iteration(s), graph-cut iteration(s), left-right post-processing, Evaluator rows."""
    from localexpstereo_amd import io as lio
    from localexpstereo_amd import stereo
    z = np.load(os.path.join(GOLDEN, "cones_crop.npz"))
    imL, imRw, gt = z["imL"], np.ascontiguousarray(z["imR_wide"]), z["gt"]
    imLw = np.concatenate([np.repeat(imL[:, :1], 64, axis=1), imL], axis=1)
    gtw = np.concatenate([np.zeros((gt.shape[0], 64), np.float32), gt], axis=1)
    e = api.HipCostVolumeEnergy.naive(imLw, imRw, max_disp=63.0, lib=lib)
    st = stereo.FastGCStereo(e, imLw, imRw, dict(lambda_=1.0), device=device, seed=3)
    st.setEvaluator(lio.Evaluator(gtw, gtw > 0, 1.0), precision=0.25)
    st.check_flow_energy = True
    ex, ra, rn = api.PROPOSE_EXPANSION, api.PROPOSE_RANSAC, api.PROPOSE_RANDOM
    tabs = [[(ex, 1), (ra, 1), (rn, 7)], [(ex, 2), (ra, 1)], [(ex, 2), (ra, 1)]]
    for u, t in zip(units, tabs):
        st.addLayer(u, t)
    lab, raw = st.run(maxIteration, (0, 1), pmInit)
    e.close()
    rows = st.log
    assert [r["index"] for r in rows] == list(range(0, pmInit + maxIteration + 2))
    assert rows[0]["all"] > 80 and rows[-1]["all"] < 25, rows
    assert st.gc_max_gap <= 1e-5
    gc_rows = rows[pmInit + 1: pmInit + maxIteration + 1]
    assert all(r["smooth"] == r["smooth"] and r["energy"] == r["data"] + r["smooth"] for r in gc_rows)
    assert lab.shape == raw.shape == imLw.shape[:2] + (4,)
    assert (lab != raw).any()                                   # post-processing replaced some labels
    return rows


def case_joint_views(lib, device, units=(16,), pmInit=1, maxIteration=1):
    """Two-view graph-cut iterations with both views in lock-step and one host team per lock-step (pm.PMRunner.gc_iteration_joint)
    give exactly the labels of the view-after-view run: the cuts of different views touch disjoint state."""
    from localexpstereo_amd import stereo
    z = np.load(os.path.join(GOLDEN, "cones_crop.npz"))
    imL, imRw = z["imL"], np.ascontiguousarray(z["imR_wide"])
    imLw = np.concatenate([np.repeat(imL[:, :1], 64, axis=1), imL], axis=1)
    ex, ra, rn = api.PROPOSE_EXPANSION, api.PROPOSE_RANSAC, api.PROPOSE_RANDOM
    tabs = [[(ex, 1), (ra, 1), (rn, 3)], [(ex, 2), (ra, 1)]]
    out = []
    for joint in (False, True):
        e = api.HipCostVolumeEnergy.naive(imLw, imRw, max_disp=63.0, lib=lib)
        st = stereo.FastGCStereo(e, imLw, imRw, dict(lambda_=1.0), device=device, seed=3)
        st.joint_views, st.concurrent_views = joint, False
        st.device_cuts = False                 # (the joint form cuts on the host: compare like with like, bit for bit)
        for u, t in zip(units, tabs):
            st.addLayer(u, t)
        lab, raw = st.run(maxIteration, (0, 1), pmInit)
        out.append((lab, raw))
        e.close()
    assert out[0][1].tobytes() == out[1][1].tobytes() and out[0][0].tobytes() == out[1][0].tobytes()


def case_expansion_graph(pr, unit=14, set_index=5, seed=41, lambda_=0.7):
    """Pairwise terms on the device (N1): the graph capacities of a lock-step computed by les_hip_batch_expansion_graph
    must be bit-identical to the host construction (liblocalexp_host.so: les_gc_build_graphs), and the moves on the
    device-built graphs must give the same labels as the host-built ones.  Cells at the image border included."""
    from localexpstereo_amd import gc as lgc
    H, W, D = pr.H, pr.W, pr.D
    rng = np.random.default_rng(seed)
    layer = om.Layer(W, H, 20, unit)
    labels = _label_map(H, W, D, seed, noise=0.05)
    lab4 = np.ascontiguousarray(labels.view(np.float32).reshape(H, W, 4))
    cur = rng.uniform(0, 0.5, (H, W)).astype(np.float32)
    prop = rng.uniform(0, 0.5, (H, W)).astype(np.float32)
    g = lgc.GraphCut(pr.e.imL, pr.e.imR, lambda_=lambda_, th_smooth=1.0, omega=10.0, epsilon=0.01)
    worst_cells = 0
    for mode, si in ((0, set_index), (1, 0), (0, len(layer.sets) - 1)):
        cells = layer.sets[si]
        regions = np.ascontiguousarray(layer.shared[cells])
        planes = random_planes(len(cells), D, H, W, seed + si, slant=0.05)
        g.labels[mode][...] = lab4
        g.costs[mode][...] = cur
        batch = api.Batch(pr.e, layer.filter[cells], regions)
        off = batch.graph_offsets()
        nn = batch.graph_nodes()
        assert nn == int(sum(int(r["w"]) * int(r["h"]) for r in regions))
        ref_payload, ref_flow0 = g.build_graphs(regions, planes, prop, off, mode=mode)
        bufs = dict(planes=api.DeviceBuffer(pr.e, max(1, len(cells)) * 16), labels=api.DeviceBuffer(pr.e, H * W * 16),
                    cur=api.DeviceBuffer(pr.e, H * W * 4), prop=api.DeviceBuffer(pr.e, H * W * 4), payload=api.DeviceBuffer(pr.e, nn * 20))
        bufs["planes"].upload(api._planes(planes).view(np.float32)); bufs["labels"].upload(lab4); bufs["cur"].upload(cur); bufs["prop"].upload(prop)
        flow0 = batch.expansion_graph(bufs["planes"].ptr, bufs["labels"].ptr, bufs["cur"].ptr, bufs["prop"].ptr, bufs["payload"].ptr, mode=mode,
                                      lambda_=lambda_, th_smooth=1.0, omega=10.0, epsilon=0.01, want_flow0=True)
        pr.e.synchronize()
        got = bufs["payload"].download((nn * 5,), np.float32)
        diff = got.view(np.uint32) != ref_payload.view(np.uint32)
        assert not diff.any(), f"graph payload differs in {int(diff.sum())} of {diff.size} values (mode {mode}, set {si})"
        assert np.allclose(flow0, ref_flow0, rtol=1e-12, atol=1e-9)
        # both against the oracle's restatement of LES/StereoEnergy.h:131-163,225-230,398-453 + LES/FastGCStereo.h:422-551 (whole-
        # matrix passes and the reference's loop order -- not the per-node fused form of the product): bit for bit, every cell
        img = pr.e.imL if mode == 0 else pr.e.imR
        for ci in range(len(cells)):
            r = regions[ci]
            o_pay, o_flow = om.expansion_graph(img, lab4, cur, prop, (int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), tuple(planes[ci]),
                                               lambda_=lambda_, th_smooth=1.0, omega=10.0, epsilon=0.01)
            lo, n = int(off[ci]) * 5, int(r["w"]) * int(r["h"]) * 5
            assert np.array_equal(got[lo:lo + n].view(np.uint32), o_pay.reshape(-1).view(np.uint32)), f"device graph != oracle (cell {ci}, mode {mode})"
            assert np.array_equal(ref_payload[lo:lo + n].view(np.uint32), o_pay.reshape(-1).view(np.uint32)), f"host graph != oracle (cell {ci}, mode {mode})"
            assert abs(float(flow0[ci]) - o_flow) <= 1e-9 * max(1.0, abs(o_flow)) and abs(float(ref_flow0[ci]) - o_flow) <= 1e-9 * max(1.0, abs(o_flow))
        assert (got.reshape(-1, 5)[:, 1:] > 0).mean() > 0.2            # the instance has real pairwise structure
        # moves on device-built graphs == moves with the host construction
        flows = g.expansion_moves_prebuilt(regions, planes, prop, got, off, flow0=flow0, mode=mode)
        lab_dev, cost_dev = g.labels[mode].copy(), g.costs[mode].copy()
        g.labels[mode][...] = lab4
        g.costs[mode][...] = cur
        gap = g.expansion_moves(regions, planes, prop, mode=mode, check=True)
        assert gap <= 1e-5
        assert np.array_equal(lab_dev.view(np.uint32), g.labels[mode].view(np.uint32)) and np.array_equal(cost_dev, g.costs[mode])
        assert (lab_dev.view(np.uint32) != lab4.view(np.uint32)).any() and (flows > 0).all()
        # device-resident form: stateless host solve -> masks -> applied on the device
        masks = np.zeros(nn, np.uint8)
        lgc.solve_prebuilt(regions, got, off, masks)
        mbuf = api.DeviceBuffer(pr.e, max(1, nn))
        mbuf.upload(masks)
        # the same cuts on the device (cells that fit a workgroup's LDS): identical masks, flow = the host solver's
        if batch.max_cell_nodes <= api.Batch.MAXFLOW_MAX_NODES:
            dm, ds, df = api.DeviceBuffer(pr.e, max(1, nn)), api.DeviceBuffer(pr.e, 4 * len(cells)), api.DeviceBuffer(pr.e, 8 * len(cells))
            batch.solve_graphs(bufs["payload"].ptr, dm.ptr, ds.ptr, df.ptr)
            pr.e.synchronize()
            assert not ds.download((len(cells),), np.int32).any(), "device max-flow hit its iteration limit"
            dev_masks = dm.download((nn,), np.uint8)
            ndiff = int(((dev_masks != 0) != (masks != 0)).sum())
            assert ndiff == 0, f"device cut differs from the host cut in {ndiff} of {nn} nodes (mode {mode}, set {si})"
            assert (dev_masks != 0).any() and not (dev_masks != 0).all()
            dev_flow = df.download((len(cells),), np.float64) + np.asarray(flow0, np.float64)
            assert np.allclose(dev_flow, flows, rtol=2e-6, atol=1e-6), f"flows differ: {np.abs(dev_flow - flows).max()}"
            for b_ in (dm, ds, df):
                b_.free()
        batch.apply_masks(bufs["planes"].ptr, mbuf.ptr, bufs["cur"].ptr, bufs["prop"].ptr, bufs["labels"].ptr)
        pr.e.synchronize()
        assert np.array_equal(bufs["labels"].download((H, W, 4), np.float32).view(np.uint32), lab_dev.view(np.uint32))
        assert np.array_equal(bufs["cur"].download((H, W), np.float32), cost_dev)
        mbuf.free()
        worst_cells = max(worst_cells, len(cells))
        batch.destroy()
        for b in bufs.values():
            b.free()
    g.close()
    return worst_cells


def case_warm_start(lib, device):
    """initCurrentFast from a given labelling (LES/FastGCStereo.h:116-130): per-pixel 1 x 1 targets."""
    from localexpstereo_amd import pm
    H, W, D = 40, 52, 12
    imL = synth.make_guide(H, W, 5)
    vol = synth.make_volume(D, H, W, 6)
    o = om.Oracle(imL, None, vol, None, windR=8, eps=1e-4, th_col=0.5, max_disp=D - 1.0)
    e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=8, eps=1e-4, th_col=0.5, max_disp=D - 1.0, lib=lib)
    labels = _label_map(H, W, D, 3, noise=0.3).view(np.float32).reshape(H, W, 4)
    labels[3, 4] = (0.0, 0.0, 500.0, 0.0)                       # an invalid label: sentinel expected
    r = pm.PMRunner(e, (8,), [[(api.PROPOSE_EXPANSION, 1)]], seed=1, device=device)
    r.init_from_labels(labels, rows_per_launch=7)
    got = r.cur.cpu().numpy()
    ref = np.zeros((H, W), np.float32)
    for y in range(H):
        for x in range(W):
            fr = (max(x - 8, 0), max(y - 8, 0), min(x + 9, W) - max(x - 8, 0), min(y + 9, H) - max(y - 8, 0))
            ref[y, x] = o.unary(fr, (x, y, 1, 1), tuple(labels[y, x]), check=True)[y, x]
    assert got[3, 4] == np.float32(1e6)
    assert np.array_equal(r.labels.cpu().numpy(), labels)
    r.close(); e.close()
    return compare_maps(got, ref)
