"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors; these properties are derived from its source
(LES/GuidedFilter.h, LES/CostVolumeEnergy.h, LES/StereoEnergy.h) and from an independent numpy
restatement of the guided-filter formula.
"""
import os

import numpy as np
import pytest

from localexpstereo_amd import synth
from tests.util import GOLDEN, box_sum, guided_filter_numpy, load_cones_crop

H, W, D = 96, 120, 16


@pytest.fixture(scope="module")
def ctx(oracle_mod):
    imL, imR = load_cones_crop()
    volL = synth.make_volume(D, H, W, seed=42)
    volR = synth.make_volume(D, H, W, seed=43)
    return oracle_mod.Oracle(imL, imR, volL, volR, windR=20, eps=1e-4, th_col=0.5), imL, imR, volL, volR


def test_count_map_N(ctx):
    """(3) N(y,x) = number of in-image pixels of the 21x21 window (LES/GuidedFilter.h:43,:69)."""
    o = ctx[0]
    N = o.stats(0)[12]
    ys, xs = np.mgrid[0:H, 0:W]
    ny = np.minimum(ys + 10, H - 1) - np.maximum(ys - 10, 0) + 1
    nx = np.minimum(xs + 10, W - 1) - np.maximum(xs - 10, 0) + 1
    assert np.array_equal(N, (ny * nx).astype(np.float64))


def test_stats_match_numpy(ctx):
    o, imL = ctx[0], ctx[1]
    st = o.stats(0)
    I = imL.astype(np.float64) * (1.0 / 255)
    N = box_sum(np.ones((H, W)), 10)
    for c in range(3):
        assert np.array_equal(st[c], I[..., c])
        np.testing.assert_allclose(st[3 + c], box_sum(I[..., c], 10) / N, rtol=1e-13)
    # inverse: Sigma * inv == identity
    S = np.empty((H, W, 3, 3))
    for a in range(3):
        for b in range(3):
            S[..., a, b] = box_sum(I[..., a] * I[..., b], 10) / N - st[3 + a] * st[3 + b] + (1e-4 if a == b else 0)
    inv = np.empty((H, W, 3, 3))
    idx = {(0, 0): 6, (0, 1): 7, (0, 2): 8, (1, 1): 9, (1, 2): 10, (2, 2): 11}
    for (a, b), k in idx.items():
        inv[..., a, b] = st[k]
        inv[..., b, a] = st[k]
    eye = np.einsum("hwab,hwbc->hwac", S, inv)
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(3), eye.shape), atol=1e-8)


def test_constant_input_is_fixed_point(ctx):
    """(1) p == k  =>  q == k (cov = 0 => a = 0, b = k; LES/GuidedFilter.h:212-220,243)."""
    o = ctx[0]
    p = np.full((H, W), 0.3125, np.float32)
    np.testing.assert_allclose(o.filter_subregion((0, 0, W, H), p), 0.3125, rtol=0, atol=2e-7)
    # sub-region: only the part margined by 2R from non-image borders is valid (LES/GuidedFilter.h:298-300)
    fr = (10, 7, 70, 60)
    q = o.filter_subregion(fr, np.full((fr[3], fr[2]), 0.3125, np.float32))
    np.testing.assert_allclose(q[20:-20, 20:-20], 0.3125, rtol=0, atol=2e-7)
    assert np.max(np.abs(q - 0.3125)) > 1e-3   # ... and it really is different outside that margin


def test_linearity(ctx):
    """(2) the filter is linear in p."""
    o = ctx[0]
    rng = np.random.default_rng(0)
    fr = (5, 3, 90, 80)
    p1 = rng.random((80, 90), dtype=np.float32)
    p2 = rng.random((80, 90), dtype=np.float32)
    q1, q2 = o.filter_subregion(fr, p1), o.filter_subregion(fr, p2)
    q12 = o.filter_subregion(fr, (0.5 * p1 + 0.25 * p2).astype(np.float32))
    np.testing.assert_allclose(q12, 0.5 * q1 + 0.25 * q2, atol=3e-6)


def test_whole_image_matches_independent_numpy(ctx):
    o, imL, imR = ctx[0], ctx[1], ctx[2]
    rng = np.random.default_rng(1)
    p = (rng.random((H, W), dtype=np.float32) * 0.5).astype(np.float32)
    for mode, im in ((0, imL), (1, imR)):
        q = o.filter_subregion((0, 0, W, H), p, mode=mode)
        ref = guided_filter_numpy(im, p, 10, 1e-4)
        np.testing.assert_allclose(q, ref, rtol=0, atol=1e-6)


def test_subregion_equals_whole_image_on_target(ctx, oracle_mod):
    """(4) sub-region result on targetRect == whole-image result for every LayerManager cell,
    including image-border cells (LES/GuidedFilter.h:298-300)."""
    o = ctx[0]
    rng = np.random.default_rng(2)
    p = (rng.random((H, W), dtype=np.float32) * 0.5).astype(np.float32)
    whole = o.filter_subregion((0, 0, W, H), p)
    layer = oracle_mod.Layer(W, H, 20, 14)
    worst = 0.0
    for r in range(0, len(layer.unit), 3):
        fr = tuple(int(v) for v in layer.filter[r])
        tr = tuple(int(v) for v in layer.shared[r])
        sub = o.filter_subregion(fr, p[fr[1]:fr[1] + fr[3], fr[0]:fr[0] + fr[2]])
        sx, sy = tr[0] - fr[0], tr[1] - fr[1]
        a = sub[sy:sy + tr[3], sx:sx + tr[2]]
        b = whole[tr[1]:tr[1] + tr[3], tr[0]:tr[0] + tr[2]]
        worst = max(worst, float(np.max(np.abs(a.astype(np.float64) - b))))
    # double-precision filter: identical up to the last float bit irrespective of summation order
    assert worst <= 6e-8


def test_gather_known_answers(ctx):
    """(5) LES/CostVolumeEnergy.h:78-96."""
    o, volL = ctx[0], ctx[3]
    fr = (0, 0, W, H)
    th = np.float32(0.5)
    for k in (0, 3, D - 2):
        raw = o.gather(fr, (0, 0, float(k), 0))
        assert np.array_equal(raw, np.minimum(volL[k], th))
    raw = o.gather(fr, (0, 0, 3.5, 0))
    assert np.array_equal(raw, np.minimum(np.float32(0.5) * volL[3] + np.float32(0.5) * volL[4], th))
    assert np.array_equal(o.gather(fr, (0, 0, -2.0, 0)), np.minimum(volL[0], th))
    assert np.array_equal(o.gather(fr, (0, 0, float(D - 1), 0)), np.minimum(volL[D - 1], th))
    assert np.array_equal(o.gather(fr, (0, 0, 1e9, 0)), np.minimum(volL[D - 1], th))
    assert np.array_equal(o.gather(fr, (0, 0, float("inf"), 0)), np.minimum(volL[D - 1], th))
    assert np.array_equal(o.gather(fr, (0, 0, float("-inf"), 0)), np.minimum(volL[0], th))
    # NaN plane => COST_FOR_INVALID truncated to th_col (:80, :96)
    assert np.all(o.gather(fr, (float("nan"), 0, 1.0, 0)) == th)


def test_gather_slanted_matches_numpy(ctx):
    o, volR = ctx[0], ctx[4]
    a, b, c = np.float32(0.07), np.float32(-0.04), np.float32(6.3)
    fr = (7, 9, 100, 70)
    raw = o.gather(fr, (a, b, c, 0), mode=1)
    ys = np.arange(fr[1], fr[1] + fr[3], dtype=np.float32)[:, None]
    xs = np.arange(fr[0], fr[0] + fr[2], dtype=np.float32)[None, :]
    d = (a * xs + (b * ys + c)).astype(np.float32)
    d0 = np.floor(d).astype(int)
    f1 = (d - np.floor(d)).astype(np.float32)
    f0 = (np.float32(1) - f1).astype(np.float32)
    yy, xx = np.broadcast_arrays(ys.astype(int), xs.astype(int))
    lo = np.clip(d0, 0, D - 2)
    C = (f0 * volR[lo, yy, xx]).astype(np.float32) + (f1 * volR[lo + 1, yy, xx]).astype(np.float32)
    C = np.where(d < 0, volR[0, yy, xx], np.where(d >= D - 1, volR[D - 1, yy, xx], C)).astype(np.float32)
    assert np.array_equal(raw, np.minimum(C, np.float32(0.5)))


def test_validity_mask(ctx):
    """(6) any of ds, ds +-5a +-5b outside [0, MAXD] => invalid (LES/StereoEnergy.h:586-605)."""
    o = ctx[0]
    pos = (10, 20, 50, 40)
    for pl in [(0.0, 0.0, 3.0, 0), (0.2, -0.1, 4.0, 0), (-0.3, 0.3, 12.0, 0), (0, 0, 15.0, 0), (0, 0, 15.01, 0),
               (0, 0, -0.01, 0), (1.0, 1.0, 0.0, 0)]:
        m = o.valid_mask(pos, pl)
        a, b, c = (np.float32(v) for v in pl[:3])
        ys = np.arange(pos[1], pos[1] + pos[3], dtype=np.float32)[:, None]
        xs = np.arange(pos[0], pos[0] + pos[2], dtype=np.float32)[None, :]
        ds = ((xs * a + ys * b).astype(np.float32) + c).astype(np.float32)
        a5, b5 = np.float32(a * 5), np.float32(b * 5)
        ok = (ds >= 0) & (ds <= D - 1)
        for sa in (1, -1):
            for sb in (1, -1):
                d = ((ds + np.float32(sa) * a5).astype(np.float32) + np.float32(sb) * b5).astype(np.float32)
                ok &= (d >= 0) & (d <= D - 1)
        assert np.array_equal(m, np.where(ok, 255, 0).astype(np.uint8))
    assert o.valid_mask((5, 5, 1, 1), (0, 0, 3.0, 0))[0, 0] == 255
    assert o.valid_mask((5, 5, 1, 1), (0, 0, 16.0, 0))[0, 0] == 0


def _gather_numpy(vol, fr, pl, th, mind, maxd):
    """LES/CostVolumeEnergy.h:70-98 restated independently of the oracle, array-wise in float32 (interpolate == 1)."""
    a, b, c = (np.float32(v) for v in pl[:3])
    Dn = vol.shape[0]
    D0 = int(-mind)
    ys = np.arange(fr[1], fr[1] + fr[3], dtype=np.float32)[:, None]
    xs = np.arange(fr[0], fr[0] + fr[2], dtype=np.float32)[None, :]
    with np.errstate(invalid="ignore", over="ignore"):
        d = (a * xs + (b * ys + c)).astype(np.float32)
        yy, xx = np.broadcast_arrays(ys.astype(int), xs.astype(int))
        bad = ~np.isfinite(d)
        dsafe = np.where(bad, np.float32(0), d)
        d0 = np.trunc(dsafe).astype(np.int64) + D0                      # int(d): truncation towards zero
        d1 = d0 + 1
        f1 = (dsafe - np.floor(dsafe)).astype(np.float32)
        f0 = (np.float32(1) - f1).astype(np.float32)
        ok = (d1 < Dn) & (d0 >= 0)
        lo, hi = np.clip(d0, 0, Dn - 1), np.clip(d1, 0, Dn - 1)
        C = ((f0 * vol[lo, yy, xx]).astype(np.float32) + (f1 * vol[hi, yy, xx]).astype(np.float32)).astype(np.float32)
        C = np.where(ok, C, np.float32(1e6))
        C = np.where(bad, np.float32(1e6), C)
        C = np.where(d >= np.float32(maxd), vol[Dn - 1, yy, xx], C)
        C = np.where(d < np.float32(mind), vol[0, yy, xx], C)
    return np.minimum(C, np.float32(th)).astype(np.float32)


def test_gather_and_validity_random_planes_match_numpy(oracle_mod):
    """Randomised form of the two tests above: 60 planes (slanted, out of range on either side, huge, infinite, NaN) on both views,
    with MIN_DISPARITY = 0 and -3: the oracle's gather and IsValiLabel against array-wise numpy restatements, bit for bit."""
    imL, imR = load_cones_crop()
    volL, volR = synth.make_volume(D, H, W, seed=52), synth.make_volume(D, H, W, seed=53)
    rng = np.random.default_rng(11)
    for mind in (0.0, -3.0):
        maxd = float(D - 1) + mind
        o = oracle_mod.Oracle(imL, imR, volL, volR, windR=20, eps=1e-4, th_col=0.5, max_disp=maxd, min_disp=mind)
        for k in range(60):
            fr = (int(rng.integers(0, 40)), int(rng.integers(0, 30)), int(rng.integers(1, 80)), int(rng.integers(1, 60)))
            pl = [rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(mind - 4, maxd + 4), 0.0]
            if k % 10 == 7:
                pl[2] = [1e9, -1e9, float("inf"), float("-inf"), float("nan"), 1e30][(k // 10) % 6]
            if k % 10 == 8:
                pl[0] = [float("nan"), 1e30, -1e30, float("inf")][(k // 10) % 4]
            mode = k & 1
            got = o.gather(fr, pl, mode=mode)
            ref = _gather_numpy(volR if mode else volL, fr, pl, 0.5, mind, maxd)
            assert np.array_equal(got, ref, equal_nan=True), (mind, k, pl)
            # IsValiLabel (LES/StereoEnergy.h:560-610): the five probes of the 11 x 11 patch corners and centre must lie in [MIN, MAX]
            a, b, c = (np.float32(v) for v in pl[:3])
            ys = np.arange(fr[1], fr[1] + fr[3], dtype=np.float32)[:, None]
            xs = np.arange(fr[0], fr[0] + fr[2], dtype=np.float32)[None, :]
            with np.errstate(invalid="ignore", over="ignore"):
                ds = ((xs * a + ys * b).astype(np.float32) + c).astype(np.float32)
                a5, b5 = np.float32(a * 5), np.float32(b * 5)
                okm = (ds >= np.float32(mind)) & (ds <= np.float32(maxd))
                for sa in (1, -1):
                    for sb in (1, -1):
                        dd = ((ds + np.float32(sa) * a5).astype(np.float32) + np.float32(sb) * b5).astype(np.float32)
                        okm &= (dd >= np.float32(mind)) & (dd <= np.float32(maxd))
            assert np.array_equal(o.valid_mask(fr, pl), np.where(okm, 255, 0).astype(np.uint8)), (mind, k, pl)


def test_unary_writes_only_target_and_marks_invalid(ctx):
    o = ctx[0]
    fr, tr = (19, 22, 82, 74), (39, 42, 42, 34)
    cm = o.unary(fr, tr, (0.3, 0.2, -20.0, 0.0))
    inside = np.zeros((H, W), bool)
    inside[tr[1]:tr[1] + tr[3], tr[0]:tr[0] + tr[2]] = True
    assert np.all(np.isnan(cm[~inside]))
    assert not np.any(np.isnan(cm[inside]))
    m = o.valid_mask(tr, (0.3, 0.2, -20.0, 0.0)).astype(bool)
    sub = cm[tr[1]:tr[1] + tr[3], tr[0]:tr[0] + tr[2]]
    assert np.all(sub[~m] == np.float32(1e6))
    assert np.all(sub[m] < 1.0)
    # without the check nothing is 1e6
    cm2 = o.unary(fr, tr, (0.3, 0.2, -20.0, 0.0), check=False)
    assert np.nanmax(cm2) < 1.0


def test_unary_equals_gather_then_filter(ctx):
    o = ctx[0]
    fr, tr, pl = (0, 0, 62, 62), (0, 0, 42, 42), (0.03, 0.01, 2.5, 0.0)
    raw = o.gather(fr, pl)
    q = o.filter_subregion(fr, raw)
    cm = o.unary(fr, tr, pl, check=False)
    assert np.array_equal(cm[:42, :42], q[:42, :42])


def test_golden_fixture_regression(ctx):
    """The committed oracle outputs (tests/golden/golden_unary.npz) still reproduce bit-for-bit."""
    o = ctx[0]
    g = np.load(os.path.join(GOLDEN, "golden_unary.npz"))
    for i in range(int(g["n"])):
        fr, tr = tuple(int(v) for v in g[f"fr{i}"]), tuple(int(v) for v in g[f"tr{i}"])
        cm = o.unary(fr, tr, tuple(float(v) for v in g[f"plane{i}"]), mode=int(g[f"mode{i}"]))
        x, y, w, h = tr
        assert np.array_equal(cm[y:y + h, x:x + w], g[f"out{i}"])


def test_float_variant_close_to_double(ctx, oracle_mod):
    """'GFfloat' (LES/CostVolumeEnergy.h:33-37): "results change slightly" (LES/main.cpp:74)."""
    o, imL, imR, volL, volR = ctx
    of = oracle_mod.Oracle(imL, imR, volL, volR, use_float=True)
    fr, tr, pl = (19, 22, 82, 74), (39, 42, 42, 34), (0.05, -0.03, 4.25, 0.0)
    a = o.unary(fr, tr, pl)[42:76, 39:81]
    b = of.unary(fr, tr, pl)[42:76, 39:81]
    assert np.max(np.abs(a - b)) < 5e-3


def test_wta_update(oracle_mod):
    import ctypes as C
    L = oracle_mod.lib()
    Wm, Hm = 20, 10
    cur = np.full((Hm, Wm), 5.0, np.float32)
    prop = np.full((Hm, Wm), 7.0, np.float32)
    prop[2:6, 3:9] = 1.0
    prop[3, 4] = 5.0          # equal cost: strict '>' keeps the current label (LES/FastGCStereo.h:57)
    labels = np.zeros((Hm, Wm), oracle_mod.PLANE_DT)
    L.les_oracle_wta_update(Wm, oracle_mod.Rect(2, 1, 10, 6), cur.ctypes.data_as(C.c_void_p),
                            prop.ctypes.data_as(C.c_void_p), labels.ctypes.data_as(C.c_void_p),
                            oracle_mod.Plane(1, 2, 3, 0))
    changed = labels["c"] == 3
    expect = np.zeros((Hm, Wm), bool)
    expect[2:6, 3:9] = True
    expect[3, 4] = False
    assert np.array_equal(changed, expect)
    assert np.all(cur[expect] == 1.0) and np.all(cur[~expect] == 5.0)
