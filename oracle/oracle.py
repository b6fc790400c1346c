"""ctypes binding of oracle/libles_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/les_oracle.h).  The product package localexpstereo_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libles_oracle.so")


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]

    def __repr__(self):
        return f"Rect({self.x},{self.y},{self.w},{self.h})"

    def tup(self):
        return (self.x, self.y, self.w, self.h)


class Plane(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("v", C.c_float)]

    def tup(self):
        return (self.a, self.b, self.c, self.v)


class Rng(C.Structure):
    _fields_ = [("state", C.c_uint64)]


RECT_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])
PLANE_DT = np.dtype([("a", "<f4"), ("b", "<f4"), ("c", "<f4"), ("v", "<f4")])


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("les_oracle.cpp", "les_oracle.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src if os.path.exists(s))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libles_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    u8p = C.POINTER(C.c_uint8)
    ip = C.POINTER(C.c_int)
    vp = C.c_void_p
    sig = {
        "les_rng_seed": (None, [C.POINTER(Rng), C.c_uint64]),
        "les_rng_next": (C.c_uint32, [C.POINTER(Rng)]),
        "les_rng_uniform_int": (C.c_int, [C.POINTER(Rng), C.c_int, C.c_int]),
        "les_rng_uniform_float": (C.c_float, [C.POINTER(Rng), C.c_float, C.c_float]),
        "les_rng_uniform_double": (C.c_double, [C.POINTER(Rng), C.c_double, C.c_double]),
        "les_plane_create": (Plane, [C.c_float] * 7),
        "les_plane_normal": (None, [C.POINTER(Plane), fp]),
        "les_plane_z": (C.c_float, [C.POINTER(Plane), C.c_float, C.c_float]),
        "les_layer_create": (vp, [C.c_int] * 4),
        "les_layer_destroy": (None, [vp]),
        "les_layer_height_blocks": (C.c_int, [vp]),
        "les_layer_width_blocks": (C.c_int, [vp]),
        "les_layer_num_cells": (C.c_int, [vp]),
        "les_layer_rects": (None, [vp, vp, vp, vp]),
        "les_layer_num_sets": (C.c_int, [vp]),
        "les_layer_set_size": (C.c_int, [vp, C.c_int]),
        "les_layer_set_cells": (None, [vp, C.c_int, ip]),
        "les_oracle_create": (vp, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_double,
                                   C.c_float, C.c_float, C.c_float, C.c_int]),
        "les_oracle_destroy": (None, [vp]),
        "les_oracle_create_naive": (vp, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float]),
        "les_oracle_pm_set": (None, [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int]),
        "les_oracle_pm_init": (None, [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int]),
        "les_oracle_get_stats": (None, [vp, C.c_int, vp]),
        "les_oracle_gather": (None, [vp, C.c_int, Rect, Plane, vp]),
        "les_oracle_filter_subregion": (None, [vp, C.c_int, Rect, vp, vp]),
        "les_oracle_valid_mask": (None, [vp, Rect, Plane, vp]),
        "les_oracle_unary_nocheck": (None, [vp, C.c_int, Rect, Rect, vp, C.c_int, Plane]),
        "les_oracle_unary": (None, [vp, C.c_int, Rect, Rect, vp, C.c_int, Plane]),
        "les_oracle_unary_batch": (None, [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int]),
        "les_oracle_aggregate_planes": (None, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int]),
        "les_oracle_wta_update": (None, [C.c_int, Rect, vp, vp, vp, Plane]),
        "les_random_unit_vector": (None, [C.POINTER(Rng), C.c_double, dp]),
        "les_create_random_label": (Plane, [C.POINTER(Rng), C.c_float, C.c_float, C.c_int, C.c_int]),
        "les_select_random_pixel": (None, [C.POINTER(Rng), Rect, ip, ip]),
        "les_expansion_proposal": (Plane, [C.POINTER(Rng), vp, C.c_int, Rect]),
        "les_random_perturbation_width": (C.c_float, [C.c_float, C.c_float, C.c_int]),
        "les_random_proposal": (Plane, [C.POINTER(Rng), vp, C.c_int, Rect, C.c_int, C.c_float, C.c_float]),
        "les_random_is_continued": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
        "les_ransac_proposal": (Plane, [C.POINTER(Rng), vp, C.c_int, Rect, C.c_int, C.c_float, C.c_float]),
        "les_ransac_sample_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double]),
        "les_oracle_solve_mx3": (None, [fp, fp, C.c_int, fp]),
        "les_fill_out_of_view": (None, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "les_convert_volume_l2r": (None, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "les_consistency_check": (None, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]),
        "les_post_process": (None, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
        "les_oracle_expansion_graph": (None, [vp, C.c_int, C.c_int, vp, vp, vp, Rect, Plane, C.c_float, C.c_float, C.c_float, C.c_float, vp, dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def as_rects(r):
    r = np.asarray(r)
    if r.dtype != RECT_DT:
        r = np.ascontiguousarray(r, np.int32).reshape(-1, 4).view(RECT_DT).reshape(-1)
    return np.ascontiguousarray(r)


def as_planes(p):
    p = np.asarray(p)
    if p.dtype != PLANE_DT:
        p = np.ascontiguousarray(p, np.float32).reshape(-1, 4).view(PLANE_DT).reshape(-1)
    return np.ascontiguousarray(p)


class Layer:
    """LayerManager::addLayer geometry for one layer (LES/LayerManager.h:44-185)."""

    def __init__(self, width, height, windR, unit):
        L = lib()
        h = L.les_layer_create(width, height, windR, unit)
        self.height_blocks = L.les_layer_height_blocks(h)
        self.width_blocks = L.les_layer_width_blocks(h)
        n = L.les_layer_num_cells(h)
        self.unit = np.zeros(n, RECT_DT)
        self.shared = np.zeros(n, RECT_DT)
        self.filter = np.zeros(n, RECT_DT)
        L.les_layer_rects(h, _ptr(self.unit), _ptr(self.shared), _ptr(self.filter))
        self.sets = []
        for s in range(L.les_layer_num_sets(h)):
            m = L.les_layer_set_size(h, s)
            cells = np.zeros(m, np.int32)
            L.les_layer_set_cells(h, s, cells.ctypes.data_as(C.POINTER(C.c_int)))
            self.sets.append(cells)
        L.les_layer_destroy(h)


class Oracle:
    """CostVolumeEnergy restatement (double guided filter by default); Oracle.naive(...) builds the
    NaiveStereoEnergy restatement (MiddV2 mode) behind the same methods."""

    @classmethod
    def naive(cls, imL, imR, max_disp, windR=20, eps=1e-4, alpha=0.9, th_col=10.0, th_grad=2.0, min_disp=0.0):
        self = cls.__new__(cls)
        self.L = lib()
        self.imL, self.imR = np.ascontiguousarray(imL, np.uint8), np.ascontiguousarray(imR, np.uint8)
        self.H, self.W = self.imL.shape[:2]
        self.D = int(max_disp) + 1
        self.max_disp, self.min_disp = float(max_disp), float(min_disp)
        self.h = self.L.les_oracle_create_naive(_ptr(self.imL), _ptr(self.imR), self.H, self.W, windR, eps, alpha, th_col, th_grad,
                                                self.max_disp, self.min_disp)
        return self

    def pm_init(self, units, states, labels, cur, mode=0, nthreads=0):
        units = as_rects(units)
        self.L.les_oracle_pm_init(self.h, mode, len(units), _ptr(units), _ptr(states), _ptr(labels), _ptr(cur), nthreads)

    def pm_set(self, units, shared, filt, states, table, labels, cur, prop, iteration, mode=0, nthreads=0):
        """table: list of (kind, K) with kind 0 Expansion / 1 Random / 2 Ransac."""
        units, shared, filt = as_rects(units), as_rects(shared), as_rects(filt)
        kinds = np.array([k for k, _ in table], np.int32)
        Ks = np.array([K for _, K in table], np.int32)
        self.L.les_oracle_pm_set(self.h, mode, len(units), _ptr(units), _ptr(shared), _ptr(filt), _ptr(states), len(table),
                                 _ptr(kinds), _ptr(Ks), _ptr(labels), _ptr(cur), _ptr(prop), iteration, nthreads)

    def __init__(self, imL, imR, volL, volR, windR=20, eps=1e-4, th_col=0.5, max_disp=None, min_disp=0.0,
                 use_float=False):
        self.L = lib()
        self.imL = np.ascontiguousarray(imL, np.uint8)
        self.imR = np.ascontiguousarray(imR, np.uint8) if imR is not None else None
        self.volL = np.ascontiguousarray(volL, np.float32) if volL is not None else None
        self.volR = np.ascontiguousarray(volR, np.float32) if volR is not None else None
        self.H, self.W = self.imL.shape[:2]
        v = self.volL if self.volL is not None else self.volR
        self.D = v.shape[0]
        self.max_disp = float(self.D - 1 if max_disp is None else max_disp)
        self.min_disp = float(min_disp)
        self.h = self.L.les_oracle_create(_ptr(self.imL), _ptr(self.imR), self.H, self.W, _ptr(self.volL),
                                          _ptr(self.volR), self.D, windR, eps, th_col, self.max_disp,
                                          self.min_disp, int(use_float))

    def __del__(self):
        try:
            self.L.les_oracle_destroy(self.h)
        except Exception:
            pass

    def stats(self, mode=0):
        out = np.zeros((13, self.H, self.W), np.float64)
        self.L.les_oracle_get_stats(self.h, mode, _ptr(out))
        return out

    def gather(self, fr, plane, mode=0):
        fr = Rect(*fr)
        raw = np.zeros((fr.h, fr.w), np.float32)
        self.L.les_oracle_gather(self.h, mode, fr, Plane(*plane), _ptr(raw))
        return raw

    def filter_subregion(self, fr, p, mode=0):
        fr = Rect(*fr)
        p = np.ascontiguousarray(p, np.float32)
        q = np.zeros_like(p)
        self.L.les_oracle_filter_subregion(self.h, mode, fr, _ptr(p), _ptr(q))
        return q

    def valid_mask(self, pos, plane):
        pos = Rect(*pos)
        m = np.zeros((pos.h, pos.w), np.uint8)
        self.L.les_oracle_valid_mask(self.h, pos, Plane(*plane), _ptr(m))
        return m

    def unary(self, fr, tr, plane, cost_map=None, mode=0, check=True):
        """Writes into an H x W cost map exactly like the reference call at LES/FastGCStereo.h:49."""
        if cost_map is None:
            cost_map = np.full((self.H, self.W), np.nan, np.float32)
        fr, tr = Rect(*fr), Rect(*tr)
        origin = cost_map.ctypes.data + 4 * (fr.y * self.W + fr.x)
        f = self.L.les_oracle_unary if check else self.L.les_oracle_unary_nocheck
        f(self.h, mode, fr, tr, C.c_void_p(origin), self.W, Plane(*plane))
        return cost_map

    def aggregate_planes(self, planes, mode=0, check=False, nthreads=0):
        planes = as_planes(planes)
        out = np.empty((len(planes), self.H, self.W), np.float32)
        self.L.les_oracle_aggregate_planes(self.h, mode, len(planes), _ptr(planes), _ptr(out), int(check), nthreads)
        return out

    def unary_batch(self, frs, trs, planes, cost_map=None, mode=0, check=True, nthreads=0):
        if cost_map is None:
            cost_map = np.full((self.H, self.W), np.nan, np.float32)
        frs, trs, planes = as_rects(frs), as_rects(trs), as_planes(planes)
        assert len(frs) == len(trs) == len(planes)
        self.L.les_oracle_unary_batch(self.h, mode, len(frs), _ptr(frs), _ptr(trs), _ptr(planes), _ptr(cost_map),
                                      int(check), nthreads)
        return cost_map


def consistency_check(dispL, dispR, threshold=1.5):
    """PMStereoBase::doConsistencyCheck (LES/PMStereoBase.h:111-144)."""
    L = lib()
    dl, dr = np.ascontiguousarray(dispL, np.float32), np.ascontiguousarray(dispR, np.float32)
    H, W = dl.shape
    fl, fr = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    L.les_consistency_check(_ptr(dl), _ptr(dr), H, W, threshold, _ptr(fl), _ptr(fr))
    return fl, fr


def post_process(labelsL, labelsR, imL, imR, windR=20, threshold=1.5, omega=10.0):
    """PMStereoBase::postProcess (LES/PMStereoBase.h:146-256) on two H x W x 4 float label maps; returns new maps."""
    L = lib()
    a = np.ascontiguousarray(labelsL, np.float32).copy()
    b = np.ascontiguousarray(labelsR, np.float32).copy()
    il, ir = np.ascontiguousarray(imL, np.uint8), np.ascontiguousarray(imR, np.uint8)
    H, W = il.shape[:2]
    assert a.shape == (H, W, 4) and b.shape == (H, W, 4)
    L.les_post_process(_ptr(a), _ptr(b), _ptr(il), _ptr(ir), H, W, windR, threshold, omega)
    return a, b


def fill_out_of_view(vol, mode):
    """fillOutOfView (LES/main.cpp:146-176), in place on a contiguous float32 [D][H][W] array."""
    assert vol.dtype == np.float32 and vol.flags.c_contiguous and vol.flags.writeable
    D, H, W = vol.shape
    lib().les_fill_out_of_view(_ptr(vol), D, H, W, mode)
    return vol


def convert_volume_l2r(vol):
    """convertVolumeL2R (LES/main.cpp:178-199)."""
    src = np.ascontiguousarray(vol, np.float32)
    dst = np.empty_like(src)
    D, H, W = src.shape
    lib().les_convert_volume_l2r(_ptr(src), _ptr(dst), D, H, W)
    return dst


def expansion_graph(img, labels, cur, prop, region, label1, lambda_=1.0, th_smooth=1.0, omega=10.0, epsilon=0.01):
    """Pairwise terms + graph construction of one expansion move (LES/StereoEnergy.h:131-163, 225-230, 398-453;
    LES/FastGCStereo.h:422-551) in the reference's own shape.  Returns (payload [N][5] float32, flow)."""
    L = lib()
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    lab = np.ascontiguousarray(labels, np.float32).reshape(H, W, 4)
    cur = np.ascontiguousarray(cur, np.float32)
    prop = np.ascontiguousarray(prop, np.float32)
    r = Rect(*region)
    out = np.zeros((r.w * r.h, 5), np.float32)
    flow = C.c_double(0.0)
    L.les_oracle_expansion_graph(_ptr(img), H, W, _ptr(lab), _ptr(cur), _ptr(prop), r, Plane(*label1), lambda_, th_smooth, omega, epsilon,
                                 _ptr(out), C.byref(flow))
    return out, flow.value
