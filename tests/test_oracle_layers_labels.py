"""Oracle pins for LayerManager geometry, Plane (item (8)),
cv::RNG restatement and the proposers (LES/Proposer.h)."""
import ctypes as C
import math

import numpy as np
import pytest


def _stats(layer):
    sizes = [len(s) for s in layer.sets]
    msh = (int(layer.shared["w"].max()), int(layer.shared["h"].max()))
    mfi = (int(layer.filter["w"].max()), int(layer.filter["h"].max()))
    return layer.width_blocks, layer.height_blocks, min(sizes), max(sizes), msh, mfi


@pytest.mark.parametrize("W,H,units,expect", [
    (450, 375, (5, 15, 25), [(90, 75, 396, 437, (15, 15), (55, 55)), (30, 25, 42, 56, (45, 45), (85, 85)),
                             (18, 15, 12, 20, (75, 75), (115, 115))]),
    (1436, 992, (14, 43, 129), [(103, 71, 425, 468, (42, 42), (82, 82)), (33, 23, 40, 54, (146, 132), (169, 169)),
                                (11, 8, 4, 6, (404, 387), (427, 427))]),
    (1500, 1000, (15, 45, 135), [(100, 67, 400, 425, (45, 45), (85, 85)), (33, 22, 40, 54, (150, 145), (175, 175)),
                                 (11, 7, 2, 6, (420, 460), (445, 480))]),
])
def test_layer_tables(oracle_mod, W, H, units, expect):
    """Cell grid, disjoint-set sizes and maximum rects of SURVEY.md section 8 (derived there from
    LES/LayerManager.h:92-182 independently of this code)."""
    for u, e in zip(units, expect):
        assert _stats(oracle_mod.Layer(W, H, 20, u)) == e


def test_layer_invariants(oracle_mod):
    W, H = 1436, 992
    for u in (14, 43, 129):
        L = oracle_mod.Layer(W, H, 20, u)
        # unit regions tile the image exactly
        cover = np.zeros((H, W), np.int32)
        for r in L.unit:
            cover[r["y"]:r["y"] + r["h"], r["x"]:r["x"] + r["w"]] += 1
        assert cover.min() == 1 and cover.max() == 1
        # every rect inside the image; unit within shared within filter; filter margin = windR unless clipped
        for un, sh, fi in zip(L.unit, L.shared, L.filter):
            for r in (un, sh, fi):
                assert r["x"] >= 0 and r["y"] >= 0 and r["x"] + r["w"] <= W and r["y"] + r["h"] <= H
            assert sh["x"] <= un["x"] and sh["x"] + sh["w"] >= un["x"] + un["w"]
            assert fi["x"] == max(0, sh["x"] - 20) and fi["x"] + fi["w"] == min(W, sh["x"] + sh["w"] + 20)
            assert fi["y"] == max(0, sh["y"] - 20) and fi["y"] + fi["h"] == min(H, sh["y"] + sh["h"] + 20)
        # shared regions of one disjoint set never overlap (LES/LayerManager.h:168-172) -> the GPU batch
        for cells in L.sets:
            cover[:] = 0
            for c in cells:
                r = L.shared[c]
                cover[r["y"]:r["y"] + r["h"], r["x"]:r["x"] + r["w"]] += 1
            assert cover.max() == 1
        assert sorted(np.concatenate(L.sets).tolist()) == list(range(len(L.unit)))


def test_rng_restatement(oracle_mod):
    """cv::RNG multiply-with-carry recurrence [recollection], checked against a python big-int model."""
    L = oracle_mod.lib()
    r = oracle_mod.Rng()
    L.les_rng_seed(C.byref(r), 12345)
    state = 12345
    for _ in range(100):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert L.les_rng_next(C.byref(r)) == state & 0xFFFFFFFF
    L.les_rng_seed(C.byref(r), 0)
    assert r.state == 0xFFFFFFFF
    L.les_rng_seed(C.byref(r), 7)
    xs = [L.les_rng_uniform_int(C.byref(r), 3, 10) for _ in range(2000)]
    assert min(xs) == 3 and max(xs) == 9
    fs = [L.les_rng_uniform_float(C.byref(r), -1.0, 2.0) for _ in range(2000)]
    assert -1.0 <= min(fs) and max(fs) < 2.0 and abs(np.mean(fs) - 0.5) < 0.1
    ds = [L.les_rng_uniform_double(C.byref(r), 0.0, 1.0) for _ in range(2000)]
    assert 0.0 <= min(ds) and max(ds) < 1.0 and abs(np.mean(ds) - 0.5) < 0.05


def test_plane_roundtrip(oracle_mod):
    """(8) CreatePlane(GetNormal(), GetZ(x,y), x, y) reproduces the plane (LES/Plane.h:14-58)."""
    L = oracle_mod.lib()
    rng = np.random.default_rng(3)
    for _ in range(200):
        a, b, c = rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0, 200)
        p = oracle_mod.Plane(a, b, c, 0)
        n = (C.c_float * 3)()
        L.les_plane_normal(C.byref(p), n)
        assert abs(n[0] ** 2 + n[1] ** 2 + n[2] ** 2 - 1) < 1e-6
        x, y = float(rng.integers(0, 1500)), float(rng.integers(0, 1000))
        z = L.les_plane_z(C.byref(p), x, y)
        q = L.les_plane_create(n[0], n[1], n[2], z, x, y, 0.0)
        assert abs(q.a - p.a) < 1e-5 * max(1, abs(p.a)) and abs(q.b - p.b) < 1e-5 * max(1, abs(p.b))
        assert abs(L.les_plane_z(C.byref(q), x, y) - z) < 2e-3


def test_random_label_and_proposers(oracle_mod):
    L = oracle_mod.lib()
    r = oracle_mod.Rng()
    L.les_rng_seed(C.byref(r), 99)
    MAXD = 255.0
    # createRandomLabel: z in [0,MAXD) at the pixel, normal within pi/3 of +z (LES/StereoEnergy.h:120-129)
    for _ in range(300):
        p = L.les_create_random_label(C.byref(r), 0.0, MAXD, 100, 50)
        z = L.les_plane_z(C.byref(p), 100.0, 50.0)
        assert -1e-2 <= z < MAXD + 1e-2
        n = (C.c_float * 3)()
        L.les_plane_normal(C.byref(p), n)
        assert n[2] >= math.cos(math.pi / 3) - 1e-5
    # perturbation width halves each step and stops below 0.1 (LES/Proposer.h:93-96,149-152)
    assert L.les_random_perturbation_width(0.0, MAXD, 0) == pytest.approx(127.5)
    assert L.les_random_perturbation_width(0.0, MAXD, 3) == pytest.approx(255.0 / 16)
    m_stop = next(m for m in range(40) if L.les_random_perturbation_width(0.0, MAXD, m) < 0.1)
    assert m_stop == 11
    assert L.les_random_is_continued(0, 7, 0, 0.0, MAXD) == 1
    assert L.les_random_is_continued(7, 7, 0, 0.0, MAXD) == 0
    assert L.les_random_is_continued(0, 7, m_stop, 0.0, MAXD) == 0
    # proposals on a label map
    Wm, Hm = 64, 48
    labels = np.zeros((Hm, Wm), oracle_mod.PLANE_DT)
    labels["a"], labels["b"] = 0.1, -0.05
    labels["c"] = 20.0 + np.arange(Wm)[None, :] * 0.0
    labels["c"][10:20, 30:40] = 77.0
    unit = oracle_mod.Rect(30, 10, 10, 10)
    lp = labels.ctypes.data_as(C.c_void_p)
    for _ in range(50):
        p = L.les_expansion_proposal(C.byref(r), lp, Wm, unit)
        assert (p.a, p.b, p.c) == (np.float32(0.1), np.float32(-0.05), 77.0)
    for m in (0, 3, 8):
        dz = L.les_random_perturbation_width(0.0, MAXD, m)
        for _ in range(50):
            p = L.les_random_proposal(C.byref(r), lp, Wm, unit, m, 0.0, MAXD)
            # the new plane passes within dz of the old disparity at SOME pixel of the unit
            ys, xs = np.mgrid[10:20, 30:40]
            znew = p.a * xs + p.b * ys + p.c
            zold = 0.1 * xs - 0.05 * ys + 77.0
            assert np.min(np.abs(znew - zold)) <= dz + 1e-2
    # RANSAC on an exactly planar unit recovers the plane (LES/Proposer.h:177-240)
    labels["c"] = 33.0
    for _ in range(5):
        p = L.les_ransac_proposal(C.byref(r), lp, Wm, unit, 500, 0.95, 1.0)
        assert abs(p.a - 0.1) < 1e-3 and abs(p.b + 0.05) < 1e-3 and abs(p.c - 33.0) < 5e-2
    assert L.les_ransac_sample_count(100, 100, 3, 0.95) == 1
    assert L.les_ransac_sample_count(50, 100, 3, 0.95) == int(math.log(0.05) / math.log(1 - (48 * 49 * 50) / (98 * 99 * 100)))


def test_ransac_solve_against_order_free_least_squares(oracle_mod):
    """cv::solve(A, b, x, DECOMP_SVD) as the oracle restates it (LES/Proposer.h:203,224), against numpy's double-precision least-squares
    / minimum-norm solve, which knows nothing of the oracle's accumulation order: three-point solves, refits on inlier sets with the
    reference's zero rows (:216), and rank-deficient systems (collinear points, a single point)."""
    import ctypes as C
    L = oracle_mod.lib()
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(5)

    def solve(A, b):
        A = np.ascontiguousarray(A, np.float32); b = np.ascontiguousarray(b, np.float32)
        x = np.zeros(3, np.float32)
        L.les_oracle_solve_mx3(A.ctypes.data_as(fp), b.ctypes.data_as(fp), len(b), x.ctypes.data_as(fp))
        return x

    worst = 0.0
    for trial in range(300):
        m = int(rng.choice([3, 3, 20, 196, 900]))
        x0, y0 = rng.integers(0, 1400), rng.integers(0, 900)
        A = np.stack([x0 + rng.integers(0, 45, m), y0 + rng.integers(0, 45, m), np.ones(m)], 1).astype(np.float32)
        plane = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(0, 255)])
        b = (A.astype(np.float64) @ plane + rng.normal(0, 0.3, m)).astype(np.float32)
        kind = trial % 4
        if kind == 1 and m > 3:                          # the reference's quirk: trailing all-zero rows
            A[m // 2:] = 0; b[m // 2:] = 0
        if kind == 2:                                    # collinear points: rank 2, minimum-norm solution
            A[:, 1] = A[:, 0] - x0 + y0
            b = (A.astype(np.float64) @ plane).astype(np.float32)
        if kind == 3 and trial % 8 == 3:                 # one distinct point: rank 1
            A[:] = A[0]; b[:] = b[0]
        got = solve(A, b)
        ref = np.linalg.lstsq(A.astype(np.float64), b.astype(np.float64), rcond=2 * 1.1920929e-07)[0]
        # compare what the plane is used for: its disparities over the points (the coefficients of an ill-conditioned system -- 45 px of
        # support at x ~ 1400 -- are not determined to 1e-5 individually by float data)
        dg, dr = A.astype(np.float64) @ got.astype(np.float64), A.astype(np.float64) @ ref
        worst = max(worst, float(np.max(np.abs(dg - dr)) / max(1.0, float(np.max(np.abs(dr))))))
    assert worst <= 1e-5, worst


def test_volume_preparation(oracle_mod):
    """LES/main.cpp:146-199 (N3): out-of-view fill and left->right volume conversion."""
    L = oracle_mod.lib()
    D, H, W = 6, 3, 12
    vol = np.arange(D * H * W, dtype=np.float32).reshape(D, H, W)
    v0 = vol.copy()
    L.les_fill_out_of_view(v0.ctypes.data_as(C.c_void_p), D, H, W, 0)
    for d in range(D):
        assert np.all(v0[d, :, :d] == vol[d, :, d:d + 1]) and np.array_equal(v0[d, :, d:], vol[d, :, d:])
    v1 = vol.copy()
    L.les_fill_out_of_view(v1.ctypes.data_as(C.c_void_p), D, H, W, 1)
    for d in range(1, D):
        assert np.all(v1[d, :, W - d:] == vol[d, :, W - d - 1:W - d]) and np.array_equal(v1[d, :, :W - d], vol[d, :, :W - d])
    dst = np.zeros_like(vol)
    L.les_convert_volume_l2r(vol.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), D, H, W)
    for d in range(D):
        assert np.array_equal(dst[d, :, :W - 1 - d], vol[d, :, d:W - 1])
        assert np.all(dst[d, :, W - 1 - d:] == vol[d, :, W - 1:W])
