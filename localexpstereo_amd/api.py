"""ctypes binding of the C ABI in include/localexp_hip.h (liblocalexp_hip.so).

This is plumbing for tests and bench.py: the product is the shared library and the C++ host adapter
(localexpstereo_amd/host/).  The library is loaded from csrc/ in-tree; if it is missing, or no HIP
device is present, everything here raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "liblocalexp_hip.so")

RECT_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4")])
PLANE_DT = np.dtype([("a", "<f4"), ("b", "<f4"), ("c", "<f4"), ("v", "<f4")])

# every symbol include/localexp_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "les_hip_create", "les_hip_create_naive", "les_hip_destroy", "les_hip_last_error", "les_hip_set_stream", "les_hip_set_thread_stream", "les_hip_synchronize",
    "les_hip_unary_one", "les_hip_unary_one_scratch", "les_hip_scratch_create", "les_hip_scratch_destroy", "les_hip_unary_batch", "les_hip_batch_create", "les_hip_batch_destroy",
    "les_hip_batch_num_jobs", "les_hip_batch_kernel_kind", "les_hip_batch_graph_nodes", "les_hip_batch_graph_offsets", "les_hip_batch_expansion_graph", "les_hip_batch_max_cell_nodes", "les_hip_batch_graph_solver_kind", "les_hip_refresh_volume", "les_hip_batch_solve_graphs", "les_hip_batch_solve_graphs_counted", "les_hip_batch_solve_graphs_tiled", "les_hip_batch_solve_graphs_tiled_stats", "les_hip_batch_tiled_workspace_bytes", "les_hip_batch_apply_masks", "les_hip_batch_run", "les_hip_batch_set_units", "les_hip_batch_propose", "les_hip_batch_wta",
    "les_hip_wta_update", "les_hip_malloc", "les_hip_free",
    "les_hip_memcpy_h2d", "les_hip_memcpy_d2h", "les_hip_memset", "les_hip_get_stats", "les_hip_strip_width", "les_hip_tiled_volume_bytes",
    "les_hip_calib_copy", "les_hip_calib_copy_wide", "les_hip_exchange_create", "les_hip_exchange_destroy", "les_hip_exchange_slot_floats",
    "les_hip_exchange_pack", "les_hip_exchange_unpack", "les_hip_exchange_tiles", "les_hip_fill_out_of_view", "les_hip_convert_volume_l2r", "les_hip_consistency_check", "les_hip_post_process",
]


PROPOSE_EXPANSION, PROPOSE_RANDOM, PROPOSE_RANSAC, PROPOSE_INIT = 0, 1, 2, 3


class LesHipError(RuntimeError):
    pass


class TiledStats(C.Structure):
    """les_hip_tiled_stats (include/localexp_hip.h)."""
    _fields_ = [("launches", C.c_int), ("unsolved", C.c_int), ("handed_cells", C.c_int), ("handed_nodes", C.c_longlong), ("host_ms", C.c_double)]


class Params(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("D", C.c_int), ("windR", C.c_int), ("eps", C.c_double),
                ("th_col", C.c_float), ("max_disparity", C.c_float), ("min_disparity", C.c_float),
                ("device", C.c_int), ("volumes_on_device", C.c_int)]


_libs = {}


def load(path=None):
    """Load the C-ABI library (default: the in-tree HIP build).  Raises if it does not exist."""
    from_env = path is None and bool(os.environ.get("LES_HIP_LIB"))
    path = os.path.abspath(path or os.environ.get("LES_HIP_LIB") or DEFAULT_LIB)     # LES_HIP_LIB: A/B builds of the same ABI
    if path in _libs:
        return _libs[path]
    # LES_HIP_LIB is a measurement switch between HIP builds.  Anything without a gfx950 code object (the CPU simulator build the
    # tests use, or a stranger's library with the same symbols) is refused, so that an environment variable can never put a
    # CPU path under the package; tests that exercise the simulator pass its path explicitly (or set LES_HIP_ALLOW_SIM=1).
    if from_env and os.path.exists(path) and os.environ.get("LES_HIP_ALLOW_SIM") != "1":
        blob = open(path, "rb").read()
        if b"gfx950" not in blob or b"les_march_kernel" not in blob:
            raise LesHipError(f"LES_HIP_LIB={path} holds no gfx950 code object of this package's kernels; refusing to load it "
                              "(set LES_HIP_ALLOW_SIM=1 only in tests of the simulator build)")
    # PyTorch-ROCm bundles its own HIP runtime.  If this library initialises the system runtime first and torch
    # initialises CUDA/HIP later in the same process, torch reports "No HIP GPUs are available".  Loading torch's
    # runtime first makes both share one copy (bench.py / pm.py use torch tensors for device memory anyway).
    if os.environ.get("LES_HIP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    if not os.path.exists(path):
        raise LesHipError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    sig = {
        "les_hip_create": (ci, [C.POINTER(vp), C.POINTER(Params), vp, vp, vp, vp]),
        "les_hip_batch_graph_nodes": (C.c_longlong, [vp]),
        "les_hip_batch_graph_offsets": (ci, [vp, vp]),
        "les_hip_batch_expansion_graph": (ci, [vp, vp, ci, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]),
        "les_hip_batch_apply_masks": (ci, [vp, vp, vp, vp, vp, vp, vp]),
        "les_hip_batch_max_cell_nodes": (C.c_longlong, [vp]),
        "les_hip_batch_graph_solver_kind": (ci, [vp]),
        "les_hip_refresh_volume": (ci, [vp, ci]),
        "les_hip_batch_solve_graphs": (ci, [vp, vp, vp, vp, vp, vp]),
        "les_hip_batch_solve_graphs_counted": (ci, [vp, vp, vp, vp, vp, vp, vp]),
        "les_hip_batch_solve_graphs_tiled": (ci, [vp, vp, vp, vp, vp, vp, vp, C.c_longlong, C.POINTER(ci), C.POINTER(ci)]),
        "les_hip_batch_solve_graphs_tiled_stats": (ci, [vp, vp, vp, vp, vp, vp, vp, C.c_longlong, C.POINTER(TiledStats)]),
        "les_hip_batch_tiled_workspace_bytes": (C.c_longlong, [vp]),
        "les_hip_calib_copy": (ci, [vp, vp, C.c_size_t, ci, vp]),
        "les_hip_calib_copy_wide": (ci, [vp, vp, C.c_size_t, ci, vp]),
        "les_hip_exchange_create": (ci, [vp, ci, ci, ci, vp, vp, C.POINTER(vp)]),
        "les_hip_exchange_destroy": (None, [vp]),
        "les_hip_exchange_slot_floats": (C.c_longlong, [vp]),
        "les_hip_exchange_pack": (ci, [vp, vp, vp, vp, vp]),
        "les_hip_exchange_unpack": (ci, [vp, vp, vp, vp, vp]),
        "les_hip_exchange_tiles": (ci, [vp, vp, vp, vp, vp]),
        "les_hip_consistency_check": (ci, [vp, vp, vp, C.c_float, vp, vp]),
        "les_hip_post_process": (ci, [vp, vp, vp, C.c_float, C.c_float]),
        "les_hip_create_naive": (ci, [C.POINTER(vp), C.POINTER(Params), vp, vp, C.c_float, C.c_float]),
        "les_hip_destroy": (None, [vp]),
        "les_hip_last_error": (C.c_char_p, []),
        "les_hip_set_stream": (ci, [vp, vp]),
        "les_hip_set_thread_stream": (ci, [vp, vp, ci]),
        "les_hip_synchronize": (ci, [vp]),
        "les_hip_unary_one": (ci, [vp, ci, vp, vp, vp, vp, ci, ci]),
        "les_hip_unary_batch": (ci, [vp, ci, ci, vp, vp, vp, vp, ci]),
        "les_hip_unary_one_scratch": (ci, [vp, vp, ci, vp, vp, vp, vp, ci, ci]),
        "les_hip_scratch_create": (ci, [vp, C.POINTER(vp)]),
        "les_hip_scratch_destroy": (None, [vp]),
        "les_hip_batch_create": (ci, [vp, ci, vp, vp, ci, C.POINTER(vp)]),
        "les_hip_batch_destroy": (None, [vp]),
        "les_hip_batch_num_jobs": (ci, [vp]),
        "les_hip_batch_kernel_kind": (ci, [vp, vp, ci]),
        "les_hip_batch_run": (ci, [vp, vp, ci, vp, ci, vp, ci]),
        "les_hip_batch_set_units": (ci, [vp, vp, vp]),
        "les_hip_batch_propose": (ci, [vp, vp, ci, ci, vp, vp, vp]),
        "les_hip_batch_wta": (ci, [vp, vp, vp, vp, vp, vp]),
        "les_hip_wta_update": (ci, [vp, ci, vp, vp, ci, vp, vp, vp]),
        "les_hip_malloc": (ci, [vp, C.POINTER(vp), C.c_size_t]),
        "les_hip_free": (ci, [vp, vp]),
        "les_hip_memcpy_h2d": (ci, [vp, vp, vp, C.c_size_t]),
        "les_hip_memcpy_d2h": (ci, [vp, vp, vp, C.c_size_t]),
        "les_hip_memset": (ci, [vp, vp, ci, C.c_size_t]),
        "les_hip_get_stats": (ci, [vp, ci, vp]),
        "les_hip_strip_width": (ci, [ci]),
        "les_hip_tiled_volume_bytes": (C.c_size_t, [vp, ci]),
        "les_hip_fill_out_of_view": (ci, [vp, ci, ci, ci, ci, ci, vp]),
        "les_hip_convert_volume_l2r": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _libs[path] = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _rects(r):
    r = np.asarray(r)
    if r.dtype != RECT_DT:
        r = np.ascontiguousarray(r, np.int32).reshape(-1, 4).view(RECT_DT).reshape(-1)
    return np.ascontiguousarray(r)


def _planes(p):
    p = np.asarray(p)
    if p.dtype != PLANE_DT:
        p = np.ascontiguousarray(p, np.float32).reshape(-1, 4).view(PLANE_DT).reshape(-1)
    return np.ascontiguousarray(p)


def fill_out_of_view(vol_dev_ptr, D, H, W, mode, device=0, stream=0, lib=None):
    """fillOutOfView (LES/main.cpp:146-176) in place on a device volume."""
    L = load(lib)
    rc = L.les_hip_fill_out_of_view(C.c_void_p(int(vol_dev_ptr)), D, H, W, mode, device, C.c_void_p(int(stream)))
    if rc:
        raise LesHipError(L.les_hip_last_error().decode())


def convert_volume_l2r(src_dev_ptr, dst_dev_ptr, D, H, W, device=0, stream=0, lib=None):
    """convertVolumeL2R (LES/main.cpp:178-199): right-view volume synthesised from the left-view one."""
    L = load(lib)
    rc = L.les_hip_convert_volume_l2r(C.c_void_p(int(src_dev_ptr)), C.c_void_p(int(dst_dev_ptr)), D, H, W, device, C.c_void_p(int(stream)))
    if rc:
        raise LesHipError(L.les_hip_last_error().decode())


class DeviceBuffer:
    """A raw device allocation made through the C ABI (used when torch is not wanted)."""

    def __init__(self, energy, nbytes):
        self.e = energy
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        energy._chk(energy.L.les_hip_malloc(energy.h, C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.e._chk(self.e.L.les_hip_memcpy_h2d(self.e.h, C.c_void_p(self.ptr), _ptr(arr), arr.nbytes))

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.e._chk(self.e.L.les_hip_memcpy_d2h(self.e.h, _ptr(out), C.c_void_p(self.ptr), out.nbytes))
        return out

    def fill(self, byte):
        self.e._chk(self.e.L.les_hip_memset(self.e.h, C.c_void_p(self.ptr), byte, self.nbytes))

    def free(self):
        if self.ptr:
            self.e.L.les_hip_free(self.e.h, C.c_void_p(self.ptr))
            self.ptr = None


class Exchange:
    """Plan of the per-set tile exchange between ranks (include/localexp_hip.h: les_hip_exchange_*): `rects_per_rank[r]` = the target
    rects of rank r's cells.  pack / unpack move this rank's tiles into / the other ranks' tiles out of the all-gather buffer; tiles()
    does pack -> ncclAllGather -> unpack inside the library on an ncclComm_t (C++ hosts); the Python driver gathers with torch.distributed."""

    def __init__(self, energy, rank, rects_per_rank):
        self.e, self.rank, self.world = energy, rank, len(rects_per_rank)
        parts = [_rects(r) for r in rects_per_rank]
        first = np.cumsum([0] + [len(p) for p in parts]).astype(np.int32)
        allr = np.concatenate(parts) if parts else np.zeros(0, RECT_DT)
        h = C.c_void_p()
        energy._chk(energy.L.les_hip_exchange_create(energy.h, rank, self.world, len(allr), _ptr(allr) if len(allr) else None, _ptr(first), C.byref(h)))
        self.h = h
        self.slot_floats = int(energy.L.les_hip_exchange_slot_floats(h))

    def pack(self, labels_ptr, cost_ptr, slot_ptr):
        self.e._chk(self.e.L.les_hip_exchange_pack(self.e.h, self.h, C.c_void_p(int(labels_ptr)), C.c_void_p(int(cost_ptr)), C.c_void_p(int(slot_ptr))))

    def unpack(self, gathered_ptr, labels_ptr, cost_ptr):
        self.e._chk(self.e.L.les_hip_exchange_unpack(self.e.h, self.h, C.c_void_p(int(gathered_ptr)), C.c_void_p(int(labels_ptr)), C.c_void_p(int(cost_ptr))))

    def tiles(self, nccl_comm, labels_ptr, cost_ptr):
        self.e._chk(self.e.L.les_hip_exchange_tiles(self.e.h, self.h, C.c_void_p(int(nccl_comm)) if nccl_comm else None, C.c_void_p(int(labels_ptr)),
                                                    C.c_void_p(int(cost_ptr))))

    def destroy(self):
        if self.h:
            self.e.L.les_hip_exchange_destroy(self.h)
            self.h = None


class Batch:
    def __init__(self, energy, filter_rects, target_rects, out_slabs=False):
        """out_slabs: False / 0 = every call writes the one H x W map; True / 1 = call i writes slab i of [n][H][W]; k > 1 = call i writes slab
        i // k (k consecutive calls -- the cells of a disjoint set -- share a map: several proposal slots of the set in one launch)."""
        self.e = energy
        self.frs = _rects(filter_rects)
        self.trs = _rects(target_rects)
        self.n = len(self.frs)
        self.out_slabs = int(out_slabs)
        h = C.c_void_p()
        energy._chk(energy.L.les_hip_batch_create(energy.h, self.n, _ptr(self.frs), _ptr(self.trs), int(out_slabs), C.byref(h)))
        self.h = h

    @property
    def num_jobs(self):
        return self.e.L.les_hip_batch_num_jobs(self.h)

    def kernel_kind(self, mode=0):
        """1: the fixed-point march kernel serves this batch, 0: the fp64 strip kernel."""
        return self.e.L.les_hip_batch_kernel_kind(self.e.h, self.h, mode)

    def run(self, planes, out_dev_ptr, mode=0, check=True, planes_on_device=False):
        """planes: host array (n,4) or a device pointer (int) when planes_on_device; out_dev_ptr: int."""
        if planes_on_device:
            pp = C.c_void_p(int(planes))
        else:
            self._planes = _planes(planes)
            assert len(self._planes) == self.n
            pp = _ptr(self._planes)
        self.e._chk(self.e.L.les_hip_batch_run(self.e.h, self.h, mode, pp, int(planes_on_device), C.c_void_p(int(out_dev_ptr)), int(check)))

    def set_units(self, unit_rects):
        self.units = _rects(unit_rects)
        assert len(self.units) == self.n
        self.e._chk(self.e.L.les_hip_batch_set_units(self.e.h, self.h, _ptr(self.units)))

    def propose(self, kind, labels_dev, rng_dev, planes_dev, m=0):
        """kind: PROPOSE_EXPANSION / _RANDOM / _RANSAC / _INIT; all pointers are device addresses (int)."""
        self.e._chk(self.e.L.les_hip_batch_propose(self.e.h, self.h, kind, m, C.c_void_p(int(labels_dev)), C.c_void_p(int(rng_dev)),
                                                   C.c_void_p(int(planes_dev))))

    def wta(self, planes_dev, cur_cost_dev, prop_cost_dev, labels_dev):
        self.e._chk(self.e.L.les_hip_batch_wta(self.e.h, self.h, C.c_void_p(int(planes_dev)), C.c_void_p(int(cur_cost_dev)),
                                               C.c_void_p(int(prop_cost_dev)), C.c_void_p(int(labels_dev))))

    # -- pairwise terms / graph capacities of the expansion moves on the device ("next" row N1)
    def graph_nodes(self):
        return int(self.e.L.les_hip_batch_graph_nodes(self.h))

    def graph_offsets(self):
        off = np.zeros(self.n, np.int64)
        self.e._chk(self.e.L.les_hip_batch_graph_offsets(self.h, _ptr(off)))
        return off

    def expansion_graph(self, planes_dev, labels_dev, cur_dev, prop_dev, payload_dev, mode=0, lambda_=1.0, th_smooth=1.0, omega=10.0, epsilon=0.01,
                        want_flow0=False):
        """Graph capacities of one lock-step (LES/FastGCStereo.h:425-551 + LES/StereoEnergy.h:398-453) into payload_dev
        (5 floats per node).  Returns the per-cell t-link flow when want_flow0."""
        f0 = np.zeros(self.n, np.float64) if want_flow0 else None
        self.e._chk(self.e.L.les_hip_batch_expansion_graph(self.e.h, self.h, mode, C.c_void_p(int(planes_dev)), C.c_void_p(int(labels_dev)),
                                                           C.c_void_p(int(cur_dev)), C.c_void_p(int(prop_dev)), lambda_, th_smooth, omega, epsilon,
                                                           C.c_void_p(int(payload_dev)), _ptr(f0)))
        return f0

    MAXFLOW_MAX_NODES = 2304          # LES_HIP_MAXFLOW_MAX_NODES

    @property
    def max_cell_nodes(self):
        return int(self.e.L.les_hip_batch_max_cell_nodes(self.h))

    @property
    def graph_solver_kind(self):
        """Which one-workgroup kernel solve_graphs would launch: 0 = les_maxflow_cell.h, 1 / 2 = les_maxflow.h, -1 = a cell above the limit."""
        return int(self.e.L.les_hip_batch_graph_solver_kind(self.h))

    def solve_graphs(self, payload_dev, masks_dev, status_dev, flows_dev=None, unsolved_total_dev=None):
        """Max-flow + segment read-out of every cell's expansion graph on the device (LES/FastGCStereo.h:553-559); cells of at most
        MAXFLOW_MAX_NODES nodes.  masks (uint8 per node), status (int32 per cell: 0 solved, 1 = cut it on the host), flows (float64 per
        cell, optional) are device pointers; unsolved_total (optional): a device int that grows by one per cell that hit the iteration limit."""
        self.e._chk(self.e.L.les_hip_batch_solve_graphs_counted(self.e.h, self.h, C.c_void_p(int(payload_dev)), C.c_void_p(int(masks_dev)), C.c_void_p(int(status_dev)),
                                                                C.c_void_p(int(flows_dev)) if flows_dev else None,
                                                                C.c_void_p(int(unsolved_total_dev)) if unsolved_total_dev else None))

    def tiled_workspace_bytes(self):
        """Device scratch les_hip_batch_solve_graphs_tiled needs for this batch (109 bytes per graph node)."""
        return int(self.e.L.les_hip_batch_tiled_workspace_bytes(self.h))

    def solve_graphs_tiled(self, payload_dev, masks_dev, status_dev, workspace_dev, workspace_bytes, flows_dev=None):
        """The same for cells of any size (the coarse layers): region-parallel push-relabel over tiles of the cells, graphs resident in
        device memory (csrc/les_maxflow_tiled.h).  workspace: 256-byte aligned device scratch of tiled_workspace_bytes().  Synchronises
        the calling thread's stream.  -> launches enqueued; self.tiled_unsolved = cells that hit the launch limit (0: every cell was cut);
        self.tiled_stats = the call's les_hip_tiled_stats (cells / nodes the host cores finished from their residual graphs, their milliseconds)."""
        st = TiledStats()
        self.e._chk(self.e.L.les_hip_batch_solve_graphs_tiled_stats(self.e.h, self.h, C.c_void_p(int(payload_dev)), C.c_void_p(int(masks_dev)), C.c_void_p(int(status_dev)),
                                                                    C.c_void_p(int(flows_dev)) if flows_dev else None, C.c_void_p(int(workspace_dev)),
                                                                    C.c_longlong(int(workspace_bytes)), C.byref(st)))
        self.tiled_unsolved = st.unsolved
        self.tiled_stats = dict(launches=st.launches, unsolved=st.unsolved, handed_cells=st.handed_cells, handed_nodes=st.handed_nodes, host_ms=st.host_ms)
        return st.launches

    def apply_masks(self, planes_dev, masks_dev, cur_dev, prop_dev, labels_dev):
        """Mask updates of a lock-step on the device (LES/FastGCStereo.h:61-62); masks in graph-node order."""
        self.e._chk(self.e.L.les_hip_batch_apply_masks(self.e.h, self.h, C.c_void_p(int(planes_dev)), C.c_void_p(int(masks_dev)), C.c_void_p(int(cur_dev)),
                                                       C.c_void_p(int(prop_dev)), C.c_void_p(int(labels_dev))))

    def destroy(self):
        if self.h:
            self.e.L.les_hip_batch_destroy(self.h)
            self.h = None


class HipCostVolumeEnergy:
    """Python mirror of CostVolumeEnergy (LES/CostVolumeEnergy.h:6-184) over the C ABI."""

    def __init__(self, imL, imR, volL, volR, windR=20, eps=1e-4, th_col=0.5, max_disp=None, min_disp=0.0,
                 device=0, volumes_on_device=False, shape=None, lib=None):
        self.L = load(lib)
        self.imL = np.ascontiguousarray(imL, np.uint8) if imL is not None else None
        self.imR = np.ascontiguousarray(imR, np.uint8) if imR is not None else None
        im = self.imL if self.imL is not None else self.imR
        self.H, self.W = im.shape[:2]
        if volumes_on_device:
            self.D = int(shape[0])
            vl, vr = (C.c_void_p(int(volL)) if volL else None), (C.c_void_p(int(volR)) if volR else None)
            self._keep = None
        else:
            self._keep = [np.ascontiguousarray(v, np.float32) if v is not None else None for v in (volL, volR)]
            v = self._keep[0] if self._keep[0] is not None else self._keep[1]
            self.D = v.shape[0]
            vl, vr = _ptr(self._keep[0]), _ptr(self._keep[1])
        self.max_disp = float(self.D - 1 if max_disp is None else max_disp)
        self.params = Params(self.H, self.W, self.D, windR, eps, th_col, self.max_disp, float(min_disp), device,
                             int(bool(volumes_on_device)))
        h = C.c_void_p()
        self.h = None
        self._chk(self.L.les_hip_create(C.byref(h), C.byref(self.params), _ptr(self.imL), _ptr(self.imR), vl, vr))
        self.h = h
        self._keep = None     # host volumes were copied to HBM

    @classmethod
    def naive(cls, imL, imR, windR=20, eps=1e-4, alpha=0.9, th_col=10.0, th_grad=2.0, max_disp=63.0, min_disp=0.0, device=0, lib=None):
        """Python mirror of NaiveStereoEnergy (LES/StereoEnergy.h:629-764; MiddV2 parameters LES/main.cpp:86-121):
        image-based matching cost, no volume.  Every method of the volume-based operator works on it."""
        self = cls.__new__(cls)
        self.L = load(lib)
        self.imL = np.ascontiguousarray(imL, np.uint8)
        self.imR = np.ascontiguousarray(imR, np.uint8)
        self.H, self.W = self.imL.shape[:2]
        self.D = 1
        self.max_disp = float(max_disp)
        self.params = Params(self.H, self.W, 1, windR, eps, th_col, self.max_disp, float(min_disp), device, 0)
        self._keep = None
        h = C.c_void_p()
        self.h = None
        self._chk(self.L.les_hip_create_naive(C.byref(h), C.byref(self.params), _ptr(self.imL), _ptr(self.imR), C.c_float(alpha), C.c_float(th_grad)))
        self.h = h
        return self

    def _chk(self, rc):
        if rc != 0:
            raise LesHipError(f"liblocalexp_hip error {rc}: {self.L.les_hip_last_error().decode()}")

    def close(self):
        if self.h:
            self.L.les_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._chk(self.L.les_hip_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    def set_thread_stream(self, stream_ptr, bind=True):
        """The calling host thread's launches on this context go to `stream_ptr` (bind=False: back to the context's stream)."""
        self._chk(self.L.les_hip_set_thread_stream(self.h, C.c_void_p(int(stream_ptr)) if stream_ptr else None, int(bool(bind))))

    def synchronize(self):
        self._chk(self.L.les_hip_synchronize(self.h))

    def strip_width(self):
        return self.L.les_hip_strip_width(self.params.windR // 2)

    def refresh_volume(self, mode=0):
        """After the caller refilled the device-resident volume of `mode` in place: cost range / fixed-point scales / tiled copy re-derived."""
        self._chk(self.L.les_hip_refresh_volume(self.h, mode))

    def tiled_volume_bytes(self, mode=0):
        """Bytes of the tiled copy of the view's volume the gather of steep planes reads (0: none, see include/localexp_hip.h)."""
        return int(self.L.les_hip_tiled_volume_bytes(self.h, mode))

    def stats(self, mode=0):
        out = np.empty((self.H, self.W, 3, 4), np.float32)
        self._chk(self.L.les_hip_get_stats(self.h, mode, _ptr(out)))
        return out

    # -- the operator (names follow the reference) ------------------------------------------------
    def ComputeUnaryPotential(self, filterRect, targetRect, costs_map, plane, mode=0, check=True):
        """costs_map is the caller's H x W float32 map; like the reference call site
        (LES/FastGCStereo.h:49) the view costs_map(filterRect) is handed to the operator."""
        assert costs_map.dtype == np.float32 and costs_map.flags.c_contiguous and costs_map.shape == (self.H, self.W)
        fr, tr = _rects([filterRect]), _rects([targetRect])
        pl = _planes([plane])
        origin = costs_map.ctypes.data + 4 * (int(fr["y"][0]) * self.W + int(fr["x"][0]))
        self._chk(self.L.les_hip_unary_one(self.h, mode, _ptr(fr), _ptr(tr), _ptr(pl), C.c_void_p(origin), self.W, int(check)))
        return costs_map

    def scratch(self):
        """A caller-owned scratch handle (struct Reusable of the reference): calls through distinct handles may run concurrently."""
        h = C.c_void_p()
        self._chk(self.L.les_hip_scratch_create(self.h, C.byref(h)))
        return h

    def scratch_free(self, h):
        self.L.les_hip_scratch_destroy(h)

    def ComputeUnaryPotentialScratch(self, scratch, filterRect, targetRect, costs_map, plane, mode=0, check=True):
        """les_hip_unary_one_scratch: the re-entrant operator (ctypes releases the GIL during the call)."""
        fr, tr = _rects([filterRect]), _rects([targetRect])
        pl = _planes([plane])
        origin = costs_map.ctypes.data + 4 * (int(fr["y"][0]) * self.W + int(fr["x"][0]))
        self._chk(self.L.les_hip_unary_one_scratch(self.h, scratch, mode, _ptr(fr), _ptr(tr), _ptr(pl), C.c_void_p(origin), self.W, int(check)))
        return costs_map

    def ComputeUnaryPotentialWithoutCheck(self, filterRect, targetRect, costs_map, plane, mode=0):
        return self.ComputeUnaryPotential(filterRect, targetRect, costs_map, plane, mode, check=False)

    def unary_batch(self, filter_rects, target_rects, planes, costs_map=None, mode=0, check=True):
        if costs_map is None:
            costs_map = np.full((self.H, self.W), np.nan, np.float32)
        frs, trs, pls = _rects(filter_rects), _rects(target_rects), _planes(planes)
        assert len(frs) == len(trs) == len(pls)
        self._chk(self.L.les_hip_unary_batch(self.h, mode, len(frs), _ptr(frs), _ptr(trs), _ptr(pls), _ptr(costs_map), int(check)))
        return costs_map

    # -- dual-view post-processing (LES/PMStereoBase.h:111-256) on device label maps -----------------
    def consistency_check(self, labelsL_dev, labelsR_dev, failL_dev, failR_dev, threshold=1.5):
        self._chk(self.L.les_hip_consistency_check(self.h, C.c_void_p(int(labelsL_dev)), C.c_void_p(int(labelsR_dev)), C.c_float(threshold),
                                                   C.c_void_p(int(failL_dev)), C.c_void_p(int(failR_dev))))

    def post_process(self, labelsL_dev, labelsR_dev, threshold=1.5, omega=10.0):
        """PMStereoBase::postProcess in place on two device label maps (FastGCStereo::run calls it with 1.5)."""
        self._chk(self.L.les_hip_post_process(self.h, C.c_void_p(int(labelsL_dev)), C.c_void_p(int(labelsR_dev)), C.c_float(threshold), C.c_float(omega)))

    def post_process_host(self, labelsL, labelsR, threshold=1.5, omega=10.0):
        """Convenience form for host label maps (H x W planes): upload, post-process on the device, download."""
        out = []
        bufs = [DeviceBuffer(self, self.H * self.W * 16) for _ in range(2)]
        try:
            for b, l in zip(bufs, (labelsL, labelsR)):
                b.upload(np.ascontiguousarray(l).view(np.float32).reshape(self.H, self.W, 4))
            self.post_process(bufs[0].ptr, bufs[1].ptr, threshold, omega)
            out = [b.download((self.H, self.W, 4), np.float32) for b in bufs]
        finally:
            for b in bufs:
                b.free()
        return out

    def wta_update(self, rects, planes, cur_cost_dev, prop_cost_dev, labels_dev, planes_on_device=False):
        rects = _rects(rects)
        if planes_on_device:
            pp = C.c_void_p(int(planes))
        else:
            self._wplanes = _planes(planes)
            pp = _ptr(self._wplanes)
        self._chk(self.L.les_hip_wta_update(self.h, len(rects), _ptr(rects), pp, int(planes_on_device),
                                            C.c_void_p(int(cur_cost_dev)), C.c_void_p(int(prop_cost_dev)), C.c_void_p(int(labels_dev))))
