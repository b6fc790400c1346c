"""ctypes binding of the host graph-cut fusion library (include/localexp_host.h, localexpstereo_amd/host/les_gc.cpp).

Replaces for Python drivers: FastGCStereo::expansionMoveBK + the BK max-flow library + the pairwise terms of
StereoEnergy (LES/FastGCStereo.h:411-597, LES/StereoEnergy.h:131-230,398-453).  Host code only."""
import ctypes as C
import os

import numpy as np

from . import api

DEFAULT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "liblocalexp_host.so")
SYMBOLS = ["les_gc_create", "les_gc_destroy", "les_gc_last_error", "les_gc_labels", "les_gc_costs", "les_gc_expansion_moves",
           "les_gc_expansion_moves_prebuilt", "les_gc_solve_prebuilt", "les_gc_solve_residual", "les_gc_build_graphs", "les_gc_smoothness_cost", "les_gc_data_cost"]
_lib = None


def load(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    sig = {
        "les_gc_create": (ci, [C.POINTER(vp), ci, ci, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float]),
        "les_gc_destroy": (None, [vp]),
        "les_gc_last_error": (C.c_char_p, []),
        "les_gc_labels": (C.POINTER(C.c_float), [vp, ci]),
        "les_gc_costs": (C.POINTER(C.c_float), [vp, ci]),
        "les_gc_expansion_moves": (ci, [vp, ci, ci, vp, vp, vp, ci, ci, C.POINTER(C.c_double)]),
        "les_gc_expansion_moves_prebuilt": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
        "les_gc_build_graphs": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp]),
        "les_gc_solve_prebuilt": (ci, [ci, vp, vp, vp, ci, vp, vp]),
        "les_gc_solve_residual": (ci, [ci, vp, vp, vp, vp, ci, ci, vp, vp]),
        "les_gc_smoothness_cost": (C.c_double, [vp, ci]),
        "les_gc_data_cost": (C.c_double, [vp, ci]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def cpu_budget():
    """CPUs this process may keep busy: the hardware threads, capped by a cgroup CPU-time quota (host/BandPool.h: cpuBudget)."""
    import os
    hw = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = period = -1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, period = int(q), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota > 0 and period > 0:
        hw = min(hw, max(1, -(-quota // period)))
    return hw


def solve_prebuilt(regions, payload, offsets, masks_out, nthreads=0, lib=None, flows_out=None):
    """Max-flow + segment readout of one lock-step of device-built graphs (stateless): fills masks_out (uint8, node order) and, when
    given, flows_out (float64 per cell: the flow through the n-links)."""
    L = load(lib)
    regions = api._rects(regions)
    assert payload.dtype == np.float32 and payload.flags.c_contiguous and masks_out.dtype == np.uint8 and masks_out.flags.c_contiguous
    assert flows_out is None or (flows_out.dtype == np.float64 and flows_out.flags.c_contiguous and len(flows_out) >= len(regions))
    offsets = np.ascontiguousarray(offsets, np.int64)
    if L.les_gc_solve_prebuilt(len(regions), api._ptr(regions), api._ptr(payload), api._ptr(offsets), nthreads, api._ptr(masks_out),
                               api._ptr(flows_out) if flows_out is not None else None):
        raise RuntimeError(L.les_gc_last_error().decode())


def solve_residual(regions, rc8, ex, offsets, masks_out, nthreads=0, solver=0, lib=None, flows_out=None):
    """The cut of every cell continued from a residual graph (8 residual capacities + one excess float per node, host/ResidualCut.h):
    what the tiled device max-flow hands over for its straggler cells.  Fills masks_out; flows_out = the flow routed here."""
    L = load(lib)
    regions = api._rects(regions)
    assert rc8.dtype == np.float32 and rc8.flags.c_contiguous and ex.dtype == np.float32 and ex.flags.c_contiguous
    assert masks_out.dtype == np.uint8 and masks_out.flags.c_contiguous
    offsets = np.ascontiguousarray(offsets, np.int64)
    if L.les_gc_solve_residual(len(regions), api._ptr(regions), api._ptr(rc8), api._ptr(ex), api._ptr(offsets), nthreads, solver, api._ptr(masks_out),
                               api._ptr(flows_out) if flows_out is not None else None):
        raise RuntimeError(L.les_gc_last_error().decode())


class GraphCut:
    """Current solution (labels + unary costs per view) and the local expansion moves on it."""

    def __init__(self, imL, imR, lambda_=20.0, th_smooth=1.0, omega=10.0, epsilon=0.01, lib=None):
        self.L = load(lib)
        self.params = dict(lambda_=float(lambda_), th_smooth=float(th_smooth), omega=float(omega), epsilon=float(epsilon))
        self.imL = np.ascontiguousarray(imL, np.uint8) if imL is not None else None
        self.imR = np.ascontiguousarray(imR, np.uint8) if imR is not None else None
        im = self.imL if self.imL is not None else self.imR
        self.H, self.W = im.shape[:2]
        h = C.c_void_p()
        if self.L.les_gc_create(C.byref(h), self.H, self.W, api._ptr(self.imL), api._ptr(self.imR), lambda_, th_smooth, omega, epsilon):
            raise RuntimeError(self.L.les_gc_last_error().decode())
        self.h = h
        # zero-copy numpy views of the context's own maps
        self.labels = [np.ctypeslib.as_array(self.L.les_gc_labels(h, m), (self.H, self.W, 4)) for m in (0, 1)]
        self.costs = [np.ctypeslib.as_array(self.L.les_gc_costs(h, m), (self.H, self.W)) for m in (0, 1)]

    def expansion_moves(self, regions, planes, proposal_cost, mode=0, nthreads=0, check=False):
        """One lock-step of a disjoint set: fuse planes[i] over regions[i] (LES/FastGCStereo.h:30-63, doGC == true).
        Returns the largest flow-vs-energy gap when check is set (LES/FastGCStereo.h:561-594), else 0."""
        regions, planes = api._rects(regions), api._planes(planes)
        pc = np.ascontiguousarray(proposal_cost, np.float32)
        assert pc.shape == (self.H, self.W) and len(regions) == len(planes)
        gap = C.c_double(0)
        if self.L.les_gc_expansion_moves(self.h, mode, len(regions), api._ptr(regions), api._ptr(planes), api._ptr(pc), nthreads, int(check), C.byref(gap)):
            raise RuntimeError(self.L.les_gc_last_error().decode())
        return gap.value

    def expansion_moves_prebuilt(self, regions, planes, proposal_cost, payload, offsets, flow0=None, mode=0, nthreads=0):
        """The same lock-step on device-built graphs (api.Batch.expansion_graph): only max-flow + mask updates on the host."""
        regions, planes = api._rects(regions), api._planes(planes)
        pc = np.ascontiguousarray(proposal_cost, np.float32)
        payload = np.ascontiguousarray(payload, np.float32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        f0 = np.ascontiguousarray(flow0, np.float64) if flow0 is not None else None
        flows = np.zeros(len(regions), np.float64)
        if self.L.les_gc_expansion_moves_prebuilt(self.h, mode, len(regions), api._ptr(regions), api._ptr(planes), api._ptr(pc), api._ptr(payload),
                                                  api._ptr(offsets), api._ptr(f0), nthreads, api._ptr(flows)):
            raise RuntimeError(self.L.les_gc_last_error().decode())
        return flows

    def build_graphs(self, regions, planes, proposal_cost, offsets, mode=0):
        """Host construction of the graph payload (parity reference of the device construction)."""
        regions, planes = api._rects(regions), api._planes(planes)
        pc = np.ascontiguousarray(proposal_cost, np.float32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        total = int(offsets[-1] + regions[-1]["w"] * regions[-1]["h"]) if len(regions) else 0
        payload = np.zeros(total * 5, np.float32)
        flow0 = np.zeros(len(regions), np.float64)
        if self.L.les_gc_build_graphs(self.h, mode, len(regions), api._ptr(regions), api._ptr(planes), api._ptr(pc), api._ptr(offsets), api._ptr(payload),
                                      api._ptr(flow0)):
            raise RuntimeError(self.L.les_gc_last_error().decode())
        return payload, flow0

    def smoothness_cost(self, mode=0):
        return self.L.les_gc_smoothness_cost(self.h, mode)

    def data_cost(self, mode=0):
        return self.L.les_gc_data_cost(self.h, mode)

    def energy(self, mode=0):
        return self.data_cost(mode) + self.smoothness_cost(mode)

    def close(self):
        if self.h:
            self.labels = self.costs = None
            self.L.les_gc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
