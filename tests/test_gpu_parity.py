"""Parity of the gfx950 build against the CPU oracle, through the C ABI (-m gpu, needs an MI355X).

Small/medium sizes: direct comparison with the oracle on seeded inputs and with the committed golden
fixture.  BASELINE full size (1500x1000, ndisp 256 volume is not needed for these properties; a 24-slice
volume keeps the host side light): size-independent properties + sampled oracle cells."""
import os

import numpy as np
import pytest

from tests import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cones(oracle_mod):
    pr = pc.cones_pair(None)
    yield pr
    pr.close()


@pytest.fixture(scope="module")
def mid(oracle_mod):
    pr = pc.synth_pair(None, 375, 450, 64)          # BASELINE config 1 shape (cones 450x375, ndisp 64)
    yield pr
    pr.close()


def test_native_library_loaded(cones):
    maps = open("/proc/self/maps").read()
    assert "liblocalexp_hip.so" in maps


def test_gpu_stats(cones):
    pc.case_stats(cones)


def test_gpu_single_calls(cones):
    assert pc.case_single_calls(cones) <= pc.TIGHT


def test_gpu_special_planes(cones):
    pc.case_special_planes(cones)


def test_gpu_golden_fixture(cones):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_unary.npz"))
    for i in range(int(g["n"])):
        fr, tr = tuple(int(v) for v in g[f"fr{i}"]), tuple(int(v) for v in g[f"tr{i}"])
        got = cones.e.ComputeUnaryPotential(fr, tr, np.full((cones.H, cones.W), np.nan, np.float32),
                                            tuple(float(v) for v in g[f"plane{i}"]), mode=int(g[f"mode{i}"]))
        x, y, w, h = tr
        ref = np.full((cones.H, cones.W), np.nan, np.float32)
        ref[y:y + h, x:x + w] = g[f"out{i}"]
        pc.compare_maps(got, ref)


def test_gpu_cell_batches_all_layers(mid):
    for unit, sets in ((5, (0, 9)), (15, (0, 5, 15)), (25, (0, 7))):      # MiddV2 layers, LES/main.cpp:300-306
        for mode in (0, 1):
            pc.case_cell_batches(mid, unit=unit, sets=sets, mode=mode)


def test_gpu_init_cells(mid):
    pc.case_init_cells(mid, unit=5)


def test_gpu_plane_slabs(mid):
    pc.case_plane_slabs(mid, n=6, mode=0)


def test_gpu_empty_and_errors(cones):
    pc.case_empty_and_errors(cones)


def test_gpu_warm_start(oracle_mod):
    pc.case_warm_start(None, "cuda")


def test_gpu_widened_abi_errors(oracle_mod):
    pc.case_widened_abi_errors(None)


def test_gpu_wta(mid):
    pc.case_wta(mid)


def test_gpu_other_radii(oracle_mod):
    # guided-filter radius = windR / 2: the strip kernel is instantiated for radii 1 .. 10, 12, 15 (csrc/les_hip.hip: kStrip), the march kernel for 2 .. 10
    # (kMarch); every radius below runs on whichever serves it and is compared with the oracle
    for windR, eps, th in ((2, 1e-2, 0.5), (4, 1e-3, 0.8), (6, 1e-4, 0.5), (8, 1e-4, 0.5), (10, 1e-4, 0.5), (12, 1e-4, 0.3),
                           (14, 1e-4, 0.5), (15, 1e-4, 0.5), (16, 1e-5, 1.5), (18, 1e-4, 0.5), (20, 1e-4, 0.5), (24, 1e-4, 0.5), (30, 1e-3, 0.5)):
        pr = pc.synth_pair(None, 90, 130, 10, windR=windR, eps=eps, th_col=th)
        try:
            layer = pc.om.Layer(pr.W, pr.H, windR, 11)
            b = pc.api.Batch(pr.e, layer.filter[layer.sets[0]], layer.shared[layer.sets[0]])
            assert b.kernel_kind(0) == (1 if 2 <= windR // 2 <= 10 else 0), windR       # radii 2 .. 10 are served by the march kernel
            b.destroy()
            for s in (0, 6):
                cells = layer.sets[s]
                planes = pc.random_planes(len(cells), pr.D, pr.H, pr.W, 4 + s)
                ref = pr.o.unary_batch(layer.filter[cells], layer.shared[cells], planes)
                got = pr.e.unary_batch(layer.filter[cells], layer.shared[cells], planes)
                pc.compare_maps(got, ref)
            # whole-image slabs: fronto-parallel planes with one and two taps and slanted planes (the other specialisations of role A, the
            # wide geometry, rings longer than the window for radii 5, 6, 8, 9)
            pc.case_plane_slabs(pr, n=5, mode=0)
        finally:
            pr.close()


def test_gpu_grouped_slots(mid):
    """Three proposal slots of a disjoint set in one launch, each into its own cost map (out_slabs = cells per slot)."""
    for unit, si in ((14, 3), (43, 1)):
        worst, kind = pc.case_grouped_slots(mid, unit=unit, set_index=si, slots=3)
        assert kind == 1


def test_gpu_repeatable(mid):
    """Same inputs -> bit-identical outputs (no atomics / order dependence)."""
    layer = pc.om.Layer(mid.W, mid.H, 20, 15)
    cells = layer.sets[2]
    planes = pc.random_planes(len(cells), mid.D, mid.H, mid.W, 77)
    a = mid.e.unary_batch(layer.filter[cells], layer.shared[cells], planes)
    b = mid.e.unary_batch(layer.filter[cells], layer.shared[cells], planes)
    assert a.tobytes() == b.tobytes()


# ---------------------------------------------------------------------------------------------------
# BASELINE full image size (1500 x 1000): size-independent properties + sampled oracle cells
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full(oracle_mod):
    # BASELINE configs[2] shape with all 256 slices resident on the device; the oracle gets host copies of the slices a test
    # touches (cut by full_oracle below), as test_max_size_volume_32bit_offsets does
    import torch
    from localexpstereo_amd import api, synth
    H, W, D = 1000, 1500, 256
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    vol = torch.rand((D, H, W), device="cuda", dtype=torch.float32, generator=gen)
    imL = synth.make_guide(H, W, 1234)
    e = api.HipCostVolumeEnergy(imL, None, vol.data_ptr(), None, volumes_on_device=True, shape=(D, H, W), max_disp=D - 1)
    pr = type("P", (), {"e": e, "H": H, "W": W, "D": D, "vol": vol, "imL": imL})()
    yield pr
    e.close()
    del vol
    torch.cuda.empty_cache()


def full_oracle(pr, lo, n):
    """Oracle over host copies of slices lo .. lo+n-1 of the full fixture's volume; planes passed to it are shifted by -lo."""
    sub = pr.vol[lo:lo + n].cpu().numpy()
    return pc.om.Oracle(pr.imL, None, sub, None, max_disp=n - 1.0)


def test_full_size_constant_volume_is_fixed_point():
    """vol == k  =>  aggregated cost == min(k, th) for every plane and pixel (guided filter of a
    constant is that constant: LES/GuidedFilter.h:212-220,243)."""
    from localexpstereo_amd import api, synth
    H, W, D = 1000, 1500, 8
    im = synth.make_guide(H, W, 1234)
    vol = np.full((D, H, W), 0.3125, np.float32)
    e = api.HipCostVolumeEnergy(im, None, vol, None)
    pr = type("P", (), {"e": e, "H": H, "W": W, "D": D})()
    planes = np.array([[0, 0, 3, 0], [0.002, -0.001, 3.3, 0], [0, 0, 100, 0]], np.float32)
    out = pc.run_slabs(pr, planes)
    assert np.max(np.abs(out - 0.3125)) <= 1e-6
    e.close()


def test_full_size_cells_equal_whole_image_and_oracle(full):
    """(a) cell-batched results == the same plane aggregated over the whole image (sub-region filter
    exactness, LES/GuidedFilter.h:298-300), for all cells of one disjoint set of every layer of the
    1500x1000 geometry; (b) sampled cells against the oracle."""
    pr = full
    plane = np.array([[0.004, -0.006, 209.25, 0.0]], np.float32)       # disparities 203.3 .. 215.2 over the image: slices 200..219
    lo, n = 200, 20
    o = full_oracle(pr, lo, n)
    plane_o = plane.copy()
    plane_o[0, 2] -= lo
    whole = pc.run_slabs(pr, plane, check=False)[0]
    for unit in (15, 45, 135):                                           # LES/main.cpp:395-397 at w = 1500
        layer = pc.om.Layer(pr.W, pr.H, 20, unit)
        cells = layer.sets[len(layer.sets) // 2]
        planes = np.repeat(plane, len(cells), axis=0)
        got = pr.e.unary_batch(layer.filter[cells], layer.shared[cells], planes, check=False)
        m = ~np.isnan(got)
        assert m.sum() == int(sum(int(r["w"]) * int(r["h"]) for r in layer.shared[cells]))
        assert np.max(np.abs(got[m] - whole[m])) <= 3e-7
        pick = cells[:: max(1, len(cells) // 6)]
        # (the validity rule depends on MAX_DISPARITY, which differs between the 256-slice context and the 20-slice oracle: no check)
        ref = o.unary_batch(layer.filter[pick], layer.shared[pick], np.repeat(plane_o, len(pick), axis=0), check=False)
        mm = ~np.isnan(ref)
        pc.compare_maps(np.where(mm, got, np.nan).astype(np.float32), ref)


def test_full_size_linearity_in_cost(full):
    """Below the truncation threshold the operator is linear in the volume: planes c=k and c=k+1 and
    the half-way plane c=k+0.5 satisfy q(k+.5) = (q(k)+q(k+1))/2."""
    pr = full
    planes = np.array([[0, 0, 4, 0], [0, 0, 5, 0], [0, 0, 4.5, 0]], np.float32)
    # costs U[0,1) truncated at 0.5 are not linear; use a context whose threshold is never reached.  (th_col = 1: the
    # fixed-point march kernel resolves (th_col - min vol) / 2^22 = 2.4e-7 of cost; a threshold far above the data, e.g. 10,
    # would coarsen that to 2.4e-6 -- see DESIGN.md "Numerics" -- while the property under test stays the same.)
    from localexpstereo_amd import api, synth
    e = api.HipCostVolumeEnergy(synth.make_guide(pr.H, pr.W, 1234), None, synth.make_volume(8, pr.H, pr.W, 42), None, th_col=1.0)
    p2 = type("P", (), {"e": e, "H": pr.H, "W": pr.W, "D": 8})()
    out = pc.run_slabs(p2, planes)
    assert np.max(np.abs(out[2] - 0.5 * (out[0] + out[1]))) <= 5e-7
    e.close()


# ---------------------------------------------------------------------------------------------------
# Which kernel serves what: the fixed-point march kernel (csrc/les_march.h) takes every LayerManager cell batch and every
# whole-image hypothesis slab; everything outside its preconditions runs the fp64 strip kernel (csrc/les_kernels.h).
# ---------------------------------------------------------------------------------------------------
def test_gpu_kernel_dispatch(mid, oracle_mod, capfd):
    from localexpstereo_amd import api, synth
    e = mid.e
    for unit in (5, 15, 25):
        layer = pc.om.Layer(mid.W, mid.H, 20, unit)
        for cells in (layer.sets[0], layer.sets[len(layer.sets) - 1]):
            b = api.Batch(e, layer.filter[cells], layer.shared[cells])
            assert b.kernel_kind(0) == 1 and b.kernel_kind(1) == 1, "LayerManager cells must run the march kernel"
            b.destroy()
    full = [(0, 0, mid.W, mid.H)] * 3
    b = api.Batch(e, full, full, out_slabs=True)
    assert b.kernel_kind(0) == 1
    b.destroy()
    # a target closer than windR to a filterRect border that is not an image border: the bound on |a| does not hold -> strip kernel
    b = api.Batch(e, [(40, 40, 120, 120)], [(45, 60, 60, 60)])
    assert b.kernel_kind(0) == 0
    b.destroy()
    # non-finite costs, a cost range far above the threshold, another radius: strip kernel
    H, W, D = 80, 120, 6
    im = synth.make_guide(H, W, 3)
    vol = synth.make_volume(D, H, W, 5)
    cells = [(0, 0, W, H)]
    # ... and the library says so on stderr, once per context and reason (a 2x slower path nobody would otherwise notice)
    quiet = os.environ.get("LES_HIP_QUIET", "0") not in ("", "0")
    for v, kw, kind, note in ((vol, {}, 1, None), (np.where(vol > 0.999, np.nan, vol).astype(np.float32), {}, 0, "NaN or infinite"), (vol - 9.0, {}, 0, "below the truncation threshold"),
                              (vol, {"windR": 8}, 1, None), (vol, {"windR": 30}, 0, "no march kernel for guided-filter radius 15")):
        capfd.readouterr()
        ee = api.HipCostVolumeEnergy(im, None, v, None, **kw)
        bb = api.Batch(ee, cells, cells)
        assert bb.kernel_kind(0) == kind
        bb2 = api.Batch(ee, cells, cells)                # a second batch of the same context: no second note
        bb2.destroy()
        bb.destroy()
        ee.close()
        err = capfd.readouterr().err
        if note is None or quiet:
            assert "strip kernel instead of the march kernel" not in err
        else:
            assert err.count("strip kernel instead of the march kernel") == 1 and note in err, err


@pytest.fixture()
def strip_only(monkeypatch):
    monkeypatch.setenv("LES_HIP_KERNEL", "strip")       # read when a context is created


def test_gpu_strip_kernel_still_matches(strip_only, oracle_mod):
    """The fp64 strip kernel remains the path for everything the march kernel declines: keep it under the same parity cases."""
    pr = pc.synth_pair(None, 200, 260, 16)
    try:
        layer = pc.om.Layer(pr.W, pr.H, 20, 15)
        b = pc.api.Batch(pr.e, layer.filter[layer.sets[0]], layer.shared[layer.sets[0]])
        assert b.kernel_kind(0) == 0
        b.destroy()
        pc.case_cell_batches(pr, unit=15, sets=(0, 5), mode=0)
        pc.case_cell_batches(pr, unit=45, sets=(1,), mode=1)
        assert pc.case_single_calls(pr) <= pc.TIGHT
        pc.case_plane_slabs(pr, n=3, mode=0)
    finally:
        pr.close()


def test_gpu_march_vs_strip_kernel(oracle_mod, monkeypatch):
    """Both kernels on the same inputs: whole-image slanted and fronto-parallel planes, both views; they agree to the sum of
    their individual distances from the oracle."""
    from localexpstereo_amd import api, synth
    H, W, D = 300, 700, 12
    imL, imR = synth.make_guide(H, W, 11), synth.make_guide(H, W, 12)
    volL, volR = synth.make_volume(D, H, W, 13), synth.make_volume(D, H, W, 14)
    planes = np.concatenate([synth.fronto_planes(D)[2:5], synth.slanted_planes(3, H, W, D - 1, seed=3)])
    outs = []
    for force in (None, "strip"):
        if force:
            monkeypatch.setenv("LES_HIP_KERNEL", force)
        e = api.HipCostVolumeEnergy(imL, imR, volL, volR)
        pr = type("P", (), {"e": e, "H": H, "W": W, "D": D})()
        outs.append([pc.run_slabs(pr, planes, mode=m, check=True) for m in (0, 1)])
        e.close()
    for m in (0, 1):
        a, b = outs[0][m], outs[1][m]
        assert np.array_equal(a == np.float32(1e6), b == np.float32(1e6))
        v = a != np.float32(1e6)
        assert np.max(np.abs(a[v] - b[v])) <= 1e-6


def test_gpu_tiled_copy_taps_equal_planar_taps(oracle_mod, monkeypatch):
    assert pc.case_tiled_taps(None, H=300, W=701, D=40, monkeypatch=monkeypatch) <= pc.TIGHT


def test_gpu_unary_one_reentrant_16_threads(mid):
    """The operator is const and is called from an OpenMP team in the reference (LES/FastGCStereo.h:30-49, one Reusable per
    thread): 16 host threads, each with its own scratch handle, evaluate disjoint cells concurrently (ctypes releases the GIL),
    several proposals per cell like a cell visit (the job table of a rect pair is built once); every result equals the batch path
    bit for bit.  A second round goes through les_hip_unary_one itself (the library's hidden per-thread scratch)."""
    import threading
    e = mid.e
    layer = pc.om.Layer(mid.W, mid.H, 20, 15)
    cells = layer.sets[3][:48]
    props = [pc.random_planes(len(cells), mid.D, mid.H, mid.W, 300 + k, slant=0.1) for k in range(3)]
    refs = [e.unary_batch(layer.filter[cells], layer.shared[cells], p, check=True) for p in props]
    for hidden in (False, True):
        outs = [np.full((mid.H, mid.W), np.nan, np.float32) for _ in props]
        errs = []

        def work(tid):
            try:
                h = None if hidden else e.scratch()
                for ci in range(tid, len(cells), 16):
                    fr = tuple(int(v) for v in layer.filter[cells[ci]])
                    tr = tuple(int(v) for v in layer.shared[cells[ci]])
                    for k, p in enumerate(props):
                        if hidden:
                            e.ComputeUnaryPotential(fr, tr, outs[k], tuple(p[ci]), mode=0, check=True)
                        else:
                            e.ComputeUnaryPotentialScratch(h, fr, tr, outs[k], tuple(p[ci]), mode=0, check=True)
                if h is not None:
                    e.scratch_free(h)
            except Exception as ex:           # pragma: no cover
                errs.append(ex)
        ts = [threading.Thread(target=work, args=(t,)) for t in range(16)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        for got, ref in zip(outs, refs):
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            m = ~np.isnan(ref)
            assert got[m].tobytes() == ref[m].tobytes(), "one-call operator differs from the batch path"


def test_gpu_proposers(mid):
    pc.case_proposers(mid, unit=15, set_index=5)


def test_gpu_ransac_adaptive_schedule(mid):
    """The device RANSAC in chunks with the reference's early stop (LES/Proposer.h:193, :229-236): planar, noisy and garbage label maps, unit regions of
    the three layers' sizes, against the oracle cell by cell (planes to float round-off, generator states exactly = the same samples consumed)."""
    far = pc.case_ransac_schedule(mid, combos=((15, 0.0), (15, 0.3), (15, 3.0), (45, 1.0), (45, 30.0), (125, 0.5), (125, 40.0)))
    print("share of cells whose RANSAC loop went past its first chunk of 16 candidates:", far)
    assert far[(15, 0.0)] == 0.0 and far[(45, 30.0)] > 0.5 and far[(125, 40.0)] > 0.5, far


def test_gpu_pm_iteration(oracle_mod):
    """PatchMatch lock-steps on the device vs the oracle (proposal generation, unary costs, WTA)."""
    pr = pc.synth_pair(None, 120, 160, 16)
    try:
        steps, worst = pc.case_pm_iteration(pr, layers_units=(10, 30), plane_exact=False)
        assert steps > 100 and worst <= pc.TIGHT
    finally:
        pr.close()


def test_gpu_volume_preparation(cones):
    pc.case_volume_preparation(cones, None)


def test_gpu_expansion_graph_on_device(cones, mid):
    pc.case_expansion_graph(cones)
    pc.case_expansion_graph(mid, unit=25, set_index=3, seed=43)


def test_gpu_graph_cut_iterations(oracle_mod):
    """Local expansion moves end to end: GPU proposals + unary costs, host graph cuts (liblocalexp_host.so)."""
    hist, gap = pc.case_quality_cones_gc(None, "cuda", pm_iters=1, gc_iters=2)
    print("cones crop PM+GC (bad1.0, data, smooth):", hist, "max flow-energy gap", gap)


def test_gpu_device_cuts_vs_host_cuts(oracle_mod):
    worst, energies = pc.case_device_cuts_vs_host_cuts(None, "cuda")
    print("device cuts vs host cuts: largest fraction of differing nodes per lock-step", worst, "energies after the iterations (host, device)", energies)


def test_gpu_device_maxflow_edge_cells(cones, mid, monkeypatch):
    """Both one-workgroup solvers: csrc/les_maxflow_cell.h (every cell fits it) and csrc/les_maxflow.h (a 48 x 48 cell in the batch, or on request)."""
    pc.case_device_maxflow_edge_cells(cones, kind=0)
    pc.case_device_maxflow_edge_cells(mid, seed=8, kind=0)
    pc.case_device_maxflow_edge_cells(cones, kind=2)
    pc.case_device_maxflow_edge_cells(mid, seed=8, kind=2)
    monkeypatch.setenv("LES_HIP_MAXFLOW_CELL_KERNEL", "0")
    pc.case_device_maxflow_edge_cells(mid, seed=9, kind=2)


def test_gpu_device_maxflow_against_independent_checkers(mid, monkeypatch):
    """les_maxflow_kernel is checked WITHOUT product code on the other side: networkx (preflow-push + residual reachability = the
    canonical cut of the reference's solver) on 50 layer-0-sized cells, and exhaustive enumeration on cells of at most 4 x 4 nodes."""
    for kind in (0, 1):                                          # csrc/les_maxflow_cell.h (the product's choice for these cells), then csrc/les_maxflow.h
        monkeypatch.setenv("LES_HIP_MAXFLOW_CELL_KERNEL", str(1 - kind))
        cells, nodes, diff = pc.case_device_maxflow_vs_networkx(mid, seed=5, ncells=50, max_side=45, kind=kind)
        print(f"device max-flow (kernel {kind}) vs networkx: {cells} cells, {nodes} nodes; float-capacity cells: {diff} nodes on ties (zero energy difference, asserted)")
        assert diff <= 2e-4 * nodes
        n = pc.case_device_maxflow_vs_brute_force(mid, seed=9, ncells=40, kind=kind)
        print(f"device max-flow (kernel {kind}) vs brute force: {n} cells of at most 4 x 4 nodes, canonical cut reproduced exactly")


@pytest.mark.parametrize("th_col", [0.5, 10.0])
def test_gpu_relative_tolerance_and_its_floor(oracle_mod, th_col):
    """1e-4 RELATIVE (north_star) holds above a floor that scales with th_col, because the kernel's error is an absolute fixed-point step: on a
    volume with exact-zero and near-zero regions the floor is measured and bounded (6e-3 th_col: 3e-3 at th_col 0.5, 0.06 at 10), the absolute error
    keeps its documented bound, and costs that are exactly 0 come out within that absolute bound of 0."""
    r = pc.case_relative_error_floor(None, th_col)
    print(f"th_col {th_col}: {r}")
    assert r["evals_below_1pct_of_th"] > 10000 and r["exact_zero_costs"] >= 0
    assert r["max_abs_err"] <= 5.3e-6 * max(1.0, th_col)                 # the bound of the fuzz sweeps
    assert r["relative_floor"] <= 6e-3 * th_col, r
    assert r["max_abs_err_on_exact_zeros"] <= 5.3e-6 * max(1.0, th_col)


def test_gpu_refresh_volume(oracle_mod):
    """les_hip_refresh_volume after an in-place refill of a device-resident volume (another cost range: other fixed-point scales, the tiled copy rebuilt)
    == a context created on the new volume, bit for bit."""
    assert pc.case_refresh_volume(None, H=200, W=300, D=24) > 0


def test_gpu_plain_build_is_bit_identical_to_the_product(oracle_mod):
    """What the CPU simulator cannot see: the inline-assembly paths of the march kernel.  libles_plain.so (build.build_hip_plain: the same sources with
    the assembly replaced by plain C++) must reproduce the product's outputs bit for bit."""
    from localexpstereo_amd import build
    if not os.path.exists(build.PLAIN_SO):
        pytest.skip("libles_plain.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    n = pc.case_plain_build_equals_product(build.PLAIN_SO)
    print(f"assembly build == plain build on {n} output arrays")


def test_gpu_tiled_device_maxflow(mid):
    """les_hip_batch_solve_graphs_tiled (cells of any size, the coarse layers' cuts): the same independent checkers as the one-workgroup kernel,
    cells of several tiles against networkx and the host solver, bit-reproducible masks, and the committed hard cells against the host solver."""
    pc.case_device_maxflow_edge_cells(mid, seed=8, tiled=True)
    cells, nodes, diff = pc.case_device_maxflow_vs_networkx(mid, seed=5, ncells=30, max_side=45, tiled=True)
    assert diff <= 2e-4 * nodes
    pc.case_device_maxflow_vs_brute_force(mid, seed=9, ncells=40, tiled=True)
    cells, nodes, ties = pc.case_tiled_maxflow_large_cells(mid)
    print(f"tiled device max-flow vs networkx: {cells} cells of several tiles, {nodes} nodes, {ties} nodes on ties in the float-capacity cells")
    assert ties <= 2e-4 * nodes
    sw = pc.case_tiled_maxflow_hard_cells(mid)
    print(f"tiled device max-flow on tests/golden/hard_cells.npz: masks equal to the host solver's, {sw} nodes switch")


def test_gpu_tiled_maxflow_handover(oracle_mod, monkeypatch):
    """Straggler cells of the tiled solver finished by the host cores from their residual graphs (csrc/les_maxflow_tiled.h: collect / pack / unpack,
    host/ResidualCut.h): cells up to 300 x 260 nodes (row bands on the host), dyadic and float capacities, the committed hard crops; both host finishers;
    cuts and flow values of the solve without hand-over."""
    pr = pc.synth_pair(None, 375, 450, 8)
    try:
        handed = pc.case_tiled_maxflow_handover(pr, monkeypatch, shapes=[(300, 260), (150, 130), (65, 31), (129, 129), (200, 45), (31, 65), (64, 30)])
        print(f"hand-over: {handed} cells finished by the host cores")
        assert handed > 0
    finally:
        pr.close()


def test_gpu_gc_sets_without_round_trips(oracle_mod, monkeypatch):
    """pm.PMRunner: the finest layer's disjoint sets enqueued without per-lock-step status reads == the per-lock-step path, bit for bit, and the roll-back of a
    set whose cuts hit the iteration limit == host cuts."""
    from localexpstereo_amd import build
    build.build_host_lib()
    done, rolled = pc.case_gc_sets_without_round_trips(None, "cuda", monkeypatch, units=(14,))
    assert done > 0 and rolled > 0


def test_gpu_exchange_pack_unpack(mid):
    assert pc.case_exchange_pack_unpack(mid) > 0


def test_gpu_exchange_tiles_over_rccl(mid):
    """les_hip_exchange_tiles on a REAL RCCL communicator created by the host (here: ctypes on librccl, one rank -- the box has one GPU):
    the library resolves ncclAllGather itself, packs, gathers and unpacks on the context's stream.  With one rank the maps must come
    back unchanged; the multi-rank semantics of pack / unpack are covered by case_exchange_pack_unpack and the world-2/4 gloo tests."""
    import ctypes as C
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not found")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        e = mid.e
        rects = np.array([(3, 5, 40, 30), (100, 200, 45, 45), (0, 0, 1, 1)], np.int32)
        x = pc.api.Exchange(e, 0, [rects])
        rng = np.random.default_rng(0)
        lab, cost = rng.normal(size=(mid.H, mid.W, 4)).astype(np.float32), rng.normal(size=(mid.H, mid.W)).astype(np.float32)
        d_lab, d_cost = pc.api.DeviceBuffer(e, lab.nbytes), pc.api.DeviceBuffer(e, cost.nbytes)
        d_lab.upload(lab); d_cost.upload(cost)
        for _ in range(3):
            x.tiles(comm.value, d_lab.ptr, d_cost.ptr)
        e.synchronize()
        assert np.array_equal(d_lab.download(lab.shape, np.float32), lab) and np.array_equal(d_cost.download(cost.shape, np.float32), cost)
        with pytest.raises(pc.api.LesHipError, match="need an ncclComm_t"):
            x2 = pc.api.Exchange(e, 0, [rects, rects[:1]])
            x2.tiles(0, d_lab.ptr, d_cost.ptr)
        d_lab.free(); d_cost.free(); x.destroy()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_gpu_device_cuts_fall_back_to_the_host(oracle_mod, monkeypatch):
    """A device max-flow that gives up (iteration limit 0) reports every cell, the lock-step is then cut on the host: the iteration
    must equal, bit for bit, the one with device cuts switched off."""
    from localexpstereo_amd import gc as lgc, pm
    imL, vol, gt = pc.cones_ad_volume()
    api = pc.api
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANDOM, 3)], [(api.PROPOSE_EXPANSION, 1)]]
    out = []
    for limit in ("0", None):
        if limit is None:
            monkeypatch.delenv("LES_HIP_MAXFLOW_MAX_ITER", raising=False)
        else:
            monkeypatch.setenv("LES_HIP_MAXFLOW_MAX_ITER", limit)
        e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=20, eps=1e-4, th_col=0.12, max_disp=63.0)
        r = pm.PMRunner(e, (14, 43), table, seed=5, device="cuda")
        g = lgc.GraphCut(imL, None, lambda_=1.0)
        r.init_labels()
        r.iteration(0)
        r.device_cuts = limit is not None
        r.begin_gc(g)
        r.gc_iteration(0)
        assert r.gc_seconds.get("cells_cut_on_device", 0) == 0
        out.append(r.labels.cpu().numpy().copy())
        r.close(); e.close(); g.close()
    assert out[0].tobytes() == out[1].tobytes()


def test_gpu_partial_host_recut(oracle_mod, monkeypatch):
    """A lock-step in which only SOME cells hit the device solver's launch limit (tiled solver, limit 12, hand-over off): the driver copies the status words
    back and cuts exactly those cells on the host, keeping the device masks of the others (advisor, round 5).  The fused energy must match the run in which
    the device solves every cell (the two are minimum cuts of the same graphs; ties aside the labellings coincide)."""
    from localexpstereo_amd import gc as lgc, pm
    imL, vol, gt = pc.cones_ad_volume()
    api = pc.api
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANDOM, 1)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
    res = []
    monkeypatch.setenv("LES_HIP_MAXFLOW_HANDOVER", "0")
    monkeypatch.setenv("LES_GC_PER_LOCKSTEP_CHECK", "1")              # (the finest layer's cuts are limited too: check them per lock-step)
    for limit in ("12", None):
        if limit is None:
            monkeypatch.delenv("LES_HIP_MAXFLOW_MAX_ITER", raising=False)
        else:
            monkeypatch.setenv("LES_HIP_MAXFLOW_MAX_ITER", limit)
        e = api.HipCostVolumeEnergy(imL, None, vol, None, windR=20, eps=1e-4, th_col=0.12, max_disp=63.0)
        r = pm.PMRunner(e, (14, 43), table, seed=5, device="cuda")
        g = lgc.GraphCut(imL, None, lambda_=1.0)
        r.init_labels()
        r.iteration(0)
        r.device_cuts = "all"
        r.begin_gc(g)
        r.gc_iteration(0)
        r.sync_gc_state()
        res.append((g.data_cost(0) + g.smoothness_cost(0), dict(r.gc_seconds), r.labels.cpu().numpy().copy()))
        r.close(); e.close(); g.close()
    (e_lim, sec_lim, lab_lim), (e_dev, sec_dev, lab_dev) = res
    assert sec_lim.get("cells_recut_on_host", 0) > 0, sec_lim
    assert sec_dev.get("cells_recut_on_host", 0) == 0 and sec_dev.get("host_cuts", 0.0) == 0.0
    assert abs(e_lim - e_dev) <= 2e-3 * abs(e_dev), (e_lim, e_dev)
    assert (lab_lim != lab_dev).any(axis=-1).mean() < 0.02


def test_gpu_stereo_driver_two_views(oracle_mod):
    rows = pc.case_stereo_driver(None, "cuda", units=(5, 15, 25), pmInit=1, maxIteration=1)
    print("FastGCStereo mirror, cones crop, two views:", rows)


def test_gpu_two_views_joint_lock_steps(oracle_mod):
    pc.case_joint_views(None, "cuda", units=(8, 24))


def test_gpu_config1_cones_end_to_end():
    """BASELINE configs[0]: MiddV2 cones 450x375, ndisp 64, NaiveStereoEnergy, pmIterations 2, then graph-cut iterations
    (2 here instead of the default 5 to bound the host time) -- the whole loop of LES/main.cpp:270-328 with the unary
    costs on the MI355X.  Evaluator: bad-0.5 px, disparities quantised to 1/4 px (LES/main.cpp:280-292)."""
    pytest.importorskip("PIL")
    import os
    from localexpstereo_amd import io as lio
    from localexpstereo_amd import stereo
    data = lio.load_data(os.path.join(os.path.dirname(__file__), "golden", "cones"), ndisp=64)
    st, lab, raw = stereo.MidV2(data, iterations=2, pmIterations=2, doDual=False)
    for r in st.log:
        print("%2d %6.2f s  E=%.0f data=%.0f smooth=%.0f  all=%.2f nonocc=%.2f" % (r["index"], r["time"], r["energy"], r["data"], r["smooth"], r["all"], r["nonocc"]))
    print("graph-cut phase seconds:", st.gc_seconds)
    assert st.log[0]["all"] > 90
    assert st.log[2]["nonocc"] < 12.0            # PatchMatch iterations alone
    assert st.log[-1]["nonocc"] < 8.0 and st.log[-1]["all"] < 16.0
    en = [r["energy"] for r in st.log[3:]]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(en, en[1:]))


# Real-data REGRESSION ANCHORS (not agreement with the reference: the oracle cannot be pinned to a run of it, and these values are this
# build's own output): the reference's MidV2 mode with its defaults (5 + 2 iterations, one view) on the four bundled Middlebury-2003 pairs.
# Expected values = profiles/round4_middv2.json (tools/middv2_all.py on the MI355X, this kernel build); the bounds are a few tenths of a
# percent wide: a change of the algorithm's behaviour, not noise, moves them.  (Round 3's kernel gave 338391.6 / 306897.6 / 263654.2 / 144899.3.)
MIDDV2_EXPECTED = {            # set: (final energy, bad-0.5 all %, nonocc %, all % after the two PatchMatch iterations)
    "cones": (338395.3, 10.45, 3.44, 11.51),
    "teddy": (306893.8, 9.22, 3.92, 12.33),
    "venus": (263643.9, 2.41, 1.38, 3.57),
    "tsukuba": (144907.0, 11.38, 10.80, 13.82),
}


@pytest.mark.parametrize("name", sorted(MIDDV2_EXPECTED))
def test_gpu_middv2_all_sets_reference_defaults(name):
    pytest.importorskip("PIL")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import middv2_all
    r = middv2_all.run_set(name)
    log = r["log"]
    e_ref, all_ref, nonocc_ref, pm_all_ref = MIDDV2_EXPECTED[name]
    print(name, r["shape"], r["seconds"], "s", [(x["index"], x["energy"], x["all"], x["nonocc"]) for x in log])
    assert len(log) == 8 and log[0]["all"] > 90                       # init, 2 PatchMatch rows, 5 graph-cut rows
    assert abs(log[2]["all"] - pm_all_ref) <= 0.5
    en = [x["energy"] for x in log[3:]]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(en, en[1:])), en   # graph-cut iterations never raise the energy
    assert abs(en[-1] - e_ref) <= 2e-3 * e_ref, (en[-1], e_ref)
    assert abs(log[-1]["all"] - all_ref) <= 0.4 and abs(log[-1]["nonocc"] - nonocc_ref) <= 0.3, (log[-1]["all"], log[-1]["nonocc"])
    assert r["seconds"] < 6.0


def test_gpu_middv2_cones_dual_as_demo_bat():
    """The reference's own shipped invocation (demo.bat: `-targetDir data/MiddV2/cones -mode MiddV2 -smooth_weight 1 -doDual 1`): both views
    optimised, then the left-right check, nearest-valid fill and weighted median of LES/PMStereoBase.h:111-256 -- the post-processing row of
    the Evaluator log.  Regression anchor (this build's output, profiles/round4_middv2.json), not agreement with the reference."""
    pytest.importorskip("PIL")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import middv2_all
    r = middv2_all.run_set("cones", dual=True)
    log = r["log"]
    print(r["seconds"], "s", [(x["index"], x["energy"], x["all"], x["nonocc"]) for x in log])
    assert len(log) == 9                                              # init, 2 PatchMatch rows, 5 graph-cut rows, the post-processed result
    assert abs(log[7]["all"] - 10.45) <= 0.4 and abs(log[7]["nonocc"] - 3.44) <= 0.3
    assert abs(log[8]["all"] - 8.65) <= 0.4 and abs(log[8]["nonocc"] - 3.13) <= 0.3, (log[8]["all"], log[8]["nonocc"])
    assert log[8]["all"] < log[7]["all"] - 1.0                        # the occluded band is what the post-processing repairs
    assert r["seconds"] < 8.0


def test_gpu_midv3_small_end_to_end(tmp_path):
    """The MidV3 front end (LES/main.cpp:330-420) on a small synthetic scene with ground truth: raw .acrt volume file ->
    device ingest (right volume synthesised) -> 1 PatchMatch + 2 graph-cut iterations, two views, post-processing."""
    import torch
    from localexpstereo_amd import io as lio
    from localexpstereo_amd import stereo, synth
    H, W, D = 120, 200, 32
    imL, imR, gt = synth.make_scene(H, W, D, seed=5)
    lio.save_cost_volume(str(tmp_path / "im0.acrt"), synth.ad_volume(imL, imR, D, "cuda").cpu().numpy())
    volL = lio.load_cost_volume(str(tmp_path / "im0.acrt"), D, H, W)
    volR = lio.load_cost_volume(str(tmp_path / "im1.acrt"), D, H, W)          # absent -> None -> convertVolumeL2R
    assert volR is None
    data = dict(imL=imL, imR=imR, dispGT=gt, nonocc=np.ones((H, W), bool), ndisp=D, gt_prec=-1.0)
    st, lab, raw = stereo.MidV3(data, volL, volR, iterations=2, pmIterations=1, doDual=True, smooth_weight=0.5, mc_threshold=0.5)
    print([(r["index"], round(r["time"], 2), round(r["energy"]), round(r["all"], 2)) for r in st.log])
    assert st.log[0]["all"] > 85 and st.log[-1]["all"] < 20
    en = [r["energy"] for r in st.log[2:4]]
    assert en[1] <= en[0] * (1 + 1e-6)
    lio.write_pfm(str(tmp_path / "disp0.pfm"), stereo.disparities(lab))
    assert np.array_equal(lio.read_pfm(str(tmp_path / "disp0.pfm")), stereo.disparities(lab))


def test_gpu_ingest_files(oracle_mod, tmp_path):
    pc.case_ingest_files(None, "cuda", tmp_path, D=40, H=64, W=333)


def test_gpu_post_process(cones, mid):
    """Dual-view post-processing (LES/PMStereoBase.h:111-256): masks and labels bit-identical to the oracle."""
    assert pc.case_post_process(cones) > 0.01
    assert pc.case_post_process(mid, seed=77) > 0.005


def test_gpu_naive_energy(oracle_mod):
    """Image-based matching cost of config 1 (NaiveStereoEnergy, LES/StereoEnergy.h:629-764) vs the oracle."""
    worst = pc.case_naive(None)
    print("naive energy max abs err", worst)
    worst = pc.case_naive(None, windR=8)
    print("naive energy (windR 8) max abs err", worst)
    worst = pc.case_naive(None, windR=14)            # radius 7: the other march-kernel instantiation
    print("naive energy (windR 14) max abs err", worst)
    os.environ["LES_HIP_KERNEL"] = "strip"           # the fp64 strip kernel of the same energy stays covered
    try:
        worst = pc.case_naive(None)
    finally:
        del os.environ["LES_HIP_KERNEL"]
    print("naive energy (strip kernel) max abs err", worst)


def test_gpu_quality_on_cones_crop_naive_energy():
    hist = pc.case_quality_cones_naive(None, "cuda", iters=2)
    print("cones crop, config-1 energy: (bad1.0 %, energy) per iteration:", hist)


def test_gpu_quality_on_cones_crop():
    hist = pc.case_quality_cones(None, "cuda", iters=3)
    print("bad1.0 %, energy per iteration:", hist)


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[1]: the Adirondack-H shape 1436 x 992, ndisp 256 (synthetic volume generated on the device; the data set
# itself is not in the container).  Layers 14 / 43 / 129 = 1 % / 3 % / 9 % of the width (LES/main.cpp:395-397).
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def adirondack(oracle_mod):
    import torch
    from localexpstereo_amd import api, synth
    H, W, D = 992, 1436, 256
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    vol = torch.rand((D, H, W), device="cuda", dtype=torch.float32, generator=gen)
    host = vol.cpu().numpy()                                            # 1.46 GB: the oracle reads the same volume
    guide = synth.make_guide(H, W, 77)
    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, volumes_on_device=True, shape=(D, H, W), max_disp=D - 1)
    o = pc.om.Oracle(guide, None, host, None, max_disp=D - 1.0)
    pr = type("P", (), {"e": e, "o": o, "H": H, "W": W, "D": D, "vol": vol, "guide": guide})()
    yield pr
    e.close()
    del vol
    torch.cuda.empty_cache()


def test_adirondack_shape_cells_of_every_layer_vs_oracle(adirondack):
    """Sampled cells of a disjoint set of every layer (image-border cells included) with random slanted planes over the full
    256-slice disparity range, D = 256 gathers, validity sentinels: device == oracle; the SURVEY section 8 geometry table holds."""
    from localexpstereo_amd import api
    pr = adirondack
    expect = {14: (103 * 71, 82), 43: (33 * 23, 169), 129: (11 * 8, 427)}
    for unit in (14, 43, 129):
        layer = pc.om.Layer(pr.W, pr.H, 20, unit)
        assert len(layer.unit) == expect[unit][0] and int(layer.filter["w"].max()) == expect[unit][1]
        for s in (0, len(layer.sets) // 2, len(layer.sets) - 1):
            cells = layer.sets[s]
            b = api.Batch(pr.e, layer.filter[cells], layer.shared[cells])
            assert b.kernel_kind(0) == 1
            b.destroy()
            pick = cells[:: max(1, len(cells) // 5)][:6]
            planes = pc.random_planes(len(pick), pr.D, pr.H, pr.W, 100 + unit + s, slant=0.15)
            got = pr.e.unary_batch(layer.filter[pick], layer.shared[pick], planes, check=True)
            ref = pr.o.unary_batch(layer.filter[pick], layer.shared[pick], planes, check=True)
            pc.compare_maps(got, ref)


def test_adirondack_shape_full_patchmatch_iteration(adirondack):
    """One full PatchMatch iteration (init + 240 lock-steps, device-resident) at the configs[1] shape: properties that do not need
    the oracle at this size -- every pixel ends with a finite cost below the invalid sentinel, the WTA update never raises a
    pixel's cost, labels reproduce their own costs through a fresh evaluation, and the run is deterministic."""
    import torch
    from localexpstereo_amd import api, pm
    pr = adirondack
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)],
             [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]                                     # LES/main.cpp:391-397
    runs = []
    for rep in range(2):
        r = pm.PMRunner(pr.e, (14, 43, 129), table, seed=9, device="cuda")
        r.init_labels()
        cur0 = r.cur.clone()
        r.iteration(0)
        labels, cur = r.labels, r.cur
        torch.cuda.synchronize()
        assert bool(torch.isfinite(cur).all()) and float(cur.max()) < 1e5
        assert bool((cur <= cur0).all()), "a winner-take-all update raised a cost"
        runs.append((labels.cpu().numpy().copy(), cur.cpu().numpy().copy()))
        r.close()
    assert runs[0][0].tobytes() == runs[1][0].tobytes() and runs[0][1].tobytes() == runs[1][1].tobytes()
    # a pixel's stored cost is the aggregated cost of its label at the time it won: re-evaluate the winning label of a few cells
    labels, cur = runs[0]
    layer = pc.om.Layer(pr.W, pr.H, 20, 14)
    cells = layer.sets[5][::60]
    hit = 0
    for c in cells:
        sh, fr = layer.shared[c], layer.filter[c]
        x, y = int(sh["x"]) + int(sh["w"]) // 2, int(sh["y"]) + int(sh["h"]) // 2
        plane = labels[y, x]
        got = pr.e.unary_batch([fr], [sh], plane[None], check=True)
        ref = pr.o.unary_batch([fr], [sh], plane[None], check=True)
        pc.compare_maps(got, ref)
        hit += int(abs(float(got[y, x]) - float(cur[y, x])) <= 2e-6)
    assert hit >= 1          # later fusions of overlapping cells may have re-assigned the pixel with the same label from another cell


@pytest.mark.parametrize("scene", ["objects", "three_surfaces"])
@pytest.mark.parametrize("dual", [False, True])
def test_adirondack_shape_midv3_end_to_end(dual, scene):
    """BASELINE configs[1] / [3] substitute: the MidV3 loop of LES/main.cpp:330-420 at 1436 x 992 x 256 on two synthetic scenes with
    ground truth (pmIterations 2, iterations 5, smooth_weight 0.5) -- "objects" (nine small objects: easy cuts) and "three_surfaces" (the C++
    host demo's scene: proposals flip most of a coarse cell, hard cuts) -- unary costs / proposals / graph capacities AND every cut on the
    MI355X (round 5: the coarse layers by the tiled solver).  The Evaluator row must improve on the PatchMatch-only row, the energy must not
    increase, no cut may have fallen back to the host, and the wall-clock is asserted against 1.3 x the measured one (north_star: 10 s)."""
    import os
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import sys
    import time
    from localexpstereo_amd import stereo
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import e2e_bench
    H, W, D = 992, 1436, 256
    imL, imR, gt, volL = e2e_bench.scene_inputs(scene, H, W, D, "cuda")
    data = dict(imL=imL, imR=imR, dispGT=gt, nonocc=np.ones((H, W), bool), ndisp=D, gt_prec=-1.0)
    t0 = time.perf_counter()
    st, lab, raw = stereo.MidV3(data, volL, None, iterations=5, pmIterations=2, doDual=dual, smooth_weight=0.5, mc_threshold=0.5,
                                error_threshold=1.0, device="cuda")
    wall = time.perf_counter() - t0
    rows = [(r["index"], round(r["time"], 2), round(r["energy"]), round(r["all"], 2), round(r["nonocc"], 2)) for r in st.log]
    print(f"Adirondack-H shape, scene {scene}, dual={dual}: wall {wall:.2f} s (optimiser {st.seconds:.2f} s), gc {st.gc_seconds}, rows (idx, t, E, all, nonocc): {rows}")
    assert st.log[0]["all"] > 90.0
    pm_row, last = st.log[2], st.log[-1]
    assert last["all"] < pm_row["all"] + 0.5 and last["all"] < 20.0
    en = [r["energy"] for r in st.log[3:8]]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(en, en[1:]))
    assert st.gc_seconds.get("host_cuts", 0.0) == 0.0 and st.gc_seconds.get("tiled_locksteps", 0) > 0, "a lock-step of the coarse layers was cut on the host"
    # north_star's orientation target is 10 s; measured on the MI355X boxes of the pool (round 5, bench.py e2e sub-record, ingest included):
    # objects 1.95-2.0 s (one view) / 6.3-6.4 s (two views + post-processing), three_surfaces 3.1-3.4 / 4.7-5.4 s over the boxes of the pool.  The bounds are 1.3 x the
    # largest of those.
    bound = {("objects", False): 2.6, ("objects", True): 8.3, ("three_surfaces", False): 4.4, ("three_surfaces", True): 7.0}[(scene, dual)]
    if not os.environ.get("LES_TEST_STRICT_TIMING"):          # default: 3 x the measured numbers (a slower box of the pool must not fail a parity run); strict: 1.3 x
        bound = min(10.0, bound * 3.0 / 1.3)
    assert wall < bound, f"Adirondack-shape run (scene {scene}, dual={dual}) took {wall:.1f} s"


def test_two_ranks_rccl_equal_one_rank(tmp_path):
    """SURVEY.md 8(e) on hardware, everything a first multi-GPU box should validate at once (needs >= 2 visible GPUs; the round-end GPU box has one:
    the test then skips):
      (a) pm.PMRunner with backend nccl (= RCCL over xGMI) and the HIP build on 2 GPUs: PatchMatch iterations AND a graph-cut iteration with every
          cut on the ranks' own GPUs reproduce the 1-rank labels and costs bit for bit;
      (b) the two-view run with the view split (stereo.FastGCStereo.run: one rank group per view, per-set all-gathers inside a group, one broadcast
          per view, post-processing replicated) on 2 -- and, with 4 GPUs, 4 -- ranks reproduces the 1-rank labelling and raw labelling bit for bit;
      (c) the C++ host: `les_host_demo ranks ... nccl` = PMStereo::runDevice with rank / world and a real ncclComm_t per rank (ncclCommInitAll, one
          host thread and one GPU per rank, les_hip_exchange_tiles) instead of the loop-back transport: every rank bit-equal to the single-rank run."""
    import subprocess
    import sys
    import torch
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("needs 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LES_HIP_QUIET="1")

    def launch(worker, world, args, port):
        cmd = ([sys.executable, worker] if world == 1 else
               [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(port), worker]) + args
        subprocess.run(cmd, check=True, timeout=900, cwd=root, env=env)
    # (a)
    worker = os.path.join(root, "tests", "dist_worker.py")
    for gc_iters in ("0", "1"):
        outs = []
        for world in (1, 2):
            out = str(tmp_path / f"a{gc_iters}_w{world}.npz")
            launch(worker, world, [out, "hip", "200", "260", "24", "1", gc_iters], 29531)
            outs.append(np.load(out))
        assert int(outs[1]["bytes_exchanged"]) > 0
        assert outs[0]["labels"].tobytes() == outs[1]["labels"].tobytes() and outs[0]["cur"].tobytes() == outs[1]["cur"].tobytes(), f"gc_iters={gc_iters}"
    # (b)
    worker = os.path.join(root, "tests", "dist_worker_dual.py")
    ref = None
    for world in [1, 2] + ([4] if ngpu >= 4 else []):
        out = str(tmp_path / f"b_w{world}.npz")
        launch(worker, world, [out, "hip", "200", "260", "24", "1", "1"], 29541 + world)
        z = np.load(out)
        if ref is None:
            ref = z
        else:
            assert ref["raw"].tobytes() == z["raw"].tobytes() and ref["lab"].tobytes() == z["lab"].tobytes(), f"two views, world {world}"
    # (c)
    demo = os.path.join(root, "localexpstereo_amd", "host", "les_host_demo")
    if os.path.exists(demo):
        r = subprocess.run([demo, "ranks", "240", "160", "32", "2", "nccl"], capture_output=True, text=True, timeout=900, env=env)
        print(r.stdout[-3000:], r.stderr[-2000:])
        assert r.returncode == 0 and "les_host_demo: OK" in r.stdout


def test_config4_size_steep_planes_tiled_taps_equal_planar_taps(monkeypatch):
    """BASELINE configs[4] shape on one GPU (3000 x 2000 x 512: beyond 32-bit element offsets): steep planes whose disparity stays inside the range
    take their taps from the tiled copy of the volume (12.3 GB; the kernel's descriptor starts at the first row a job gathers -- round 5), and the
    aggregated costs are bit-identical to the taps from [D][H][W] (context created with LES_HIP_TILED=0)."""
    import torch
    from localexpstereo_amd import api, synth
    H, W, D = 2000, 3000, 512
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2**30:
        pytest.skip("not enough free HBM")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    vol = torch.empty((D, H, W), device="cuda", dtype=torch.float32)
    for d0 in range(0, D, 64):
        vol[d0:d0 + 64].uniform_(0.0, 1.0, generator=gen)
    guide = synth.make_guide(H, W, 99)
    rng = np.random.default_rng(3)
    n = 12
    planes = np.zeros((n, 4), np.float32)
    planes[:, 0] = rng.uniform(0.03, 0.07, n) * np.where(np.arange(n) % 2, 1.0, -1.0)
    planes[:, 1] = rng.uniform(-0.004, 0.004, n)
    planes[:, 2] = 256.0 - planes[:, 0] * (W / 2) - planes[:, 1] * (H / 2)
    full = [(0, 0, W, H)] * n
    outs = {}
    for tag in ("tiled", "planar"):
        if tag == "planar":
            monkeypatch.setenv("LES_HIP_TILED", "0")
        else:
            monkeypatch.delenv("LES_HIP_TILED", raising=False)
        e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, volumes_on_device=True, shape=(D, H, W), max_disp=D - 1)
        assert (e.tiled_volume_bytes(0) > 0) == (tag == "tiled")
        b = api.Batch(e, full, full, out_slabs=True)
        assert b.kernel_kind(0) == 1
        out = torch.zeros((n, H, W), device="cuda", dtype=torch.float32)
        b.run(planes, out.data_ptr(), mode=0, check=True)
        e.synchronize()
        outs[tag] = out
        b.destroy(); e.close()
    assert torch.equal(outs["tiled"], outs["planar"])
    v = outs["tiled"][outs["tiled"] != 1e6]
    assert 0.2 < float(v.mean()) < 0.5                       # (U[0,1) costs truncated at 0.5 and averaged)


def test_max_size_volume_32bit_offsets(oracle_mod):
    """BASELINE configs[4] shape: 3000 x 2000 x 512 (12.3 GB, 3.07e9 floats: element offsets above 2^31).
    Fronto-parallel planes in the lowest and the highest slices are checked against the oracle on host copies of
    just those slices (same arithmetic: the lerp fraction is exact); then STEEP planes (tiled-copy taps) in slices 416-443 and 484-511."""
    import torch
    from localexpstereo_amd import api, synth
    H, W, D = 2000, 3000, 512
    free, _ = torch.cuda.mem_get_info()
    if free < 32 * 2**30:
        pytest.skip("not enough free HBM")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    vol = torch.rand((D, H, W), device="cuda", dtype=torch.float32, generator=gen)
    guide = synth.make_guide(H, W, 99)
    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, volumes_on_device=True, shape=(D, H, W), max_disp=D - 1)
    pr = type("P", (), {"e": e, "H": H, "W": W, "D": D})()
    planes = np.array([[0, 0, 0.5, 0], [0, 0, 510.5, 0]], np.float32)
    layer = pc.om.Layer(W, H, 20, 30)                                     # int(w * 0.01) at w = 3000
    cells = layer.sets[7][::97]
    for pl, lo in ((planes[0], 0), (planes[1], 509)):
        sub = vol[lo:lo + 3].cpu().numpy()
        o = pc.om.Oracle(guide, None, sub, None, max_disp=2.0)
        pl_o = pl.copy()
        pl_o[2] -= lo
        n = len(cells)
        got = e.unary_batch(layer.filter[cells], layer.shared[cells], np.repeat(pl[None], n, 0), check=False)
        ref = o.unary_batch(layer.filter[cells], layer.shared[cells], np.repeat(pl_o[None], n, 0), check=False)
        pc.compare_maps(got, ref)
    # STEEP planes against the oracle where the element offsets exceed 2^31 (round 6; LES/CostVolumeEnergy.h:70-98 is what the taps must equal): |a| = 5/32
    # disparities per column puts the cell batches on the short gather FROM THE TILED COPY (|a| >= 0.125 in the two-job geometry; the copy is 12.3 GB, its
    # descriptor starts at the job's own rows).  The cells of one column of the grid share their disparity range, so the oracle runs on host copies of the 28
    # slices that column touches: slices 416 ... 443 (element offsets 2.5e9 ... 2.7e9), and 484 ... 511 with planes that leave the range at the top
    # (the d >= MAXD branch, :79, next to the last interpolated pair).
    assert e.tiled_volume_bytes(0) > 0, "the tiled copy of the volume was not built (needs twice the volume + 4 GB of free memory)"
    ux = layer.unit["x"]
    col = np.array([c for c in range(len(layer.unit)) if 1440 <= ux[c] < 1470], np.int64)[::5]      # one column of cells in the middle of the image, every fifth row
    assert len(col) >= 8
    fr, sh = layer.filter[col], layer.shared[col]
    x_lo, x_hi = int(fr["x"].min()), int((fr["x"] + fr["w"]).max())
    checked = 0
    for lo, S, top in ((416, 28, False), (484, 28, True)):
        sub = vol[lo:lo + S].cpu().numpy()
        o = pc.om.Oracle(guide, None, sub, None, max_disp=float(S - 1))
        # (dyadic coefficients: a = +-5/32, b = 2^-11 and c a multiple of 2^-11 make d = a*x + (b*y + c) EXACT in float for the device's plane and for the
        #  oracle's plane shifted by -lo alike, so both interpolate with the same fraction -- the comparison is then as tight as at small sizes)
        for a in (0.15625, -0.15625):
            b_ = 2.0 ** -11
            span = abs(a) * (x_hi - x_lo) + b_ * H                          # disparity range of the plane over the column's filter rects
            assert span < S - 3
            d_min = (lo + S - 4.0 - 0.6 * span) if top else (lo + 1.5)       # top: the plane's upper part lies above MAXD = 511
            c0 = np.round((d_min - min(a * x_lo, a * x_hi)) * 2048.0) / 2048.0
            pl = np.array([a, b_, c0, 0], np.float32)
            pl_o = pl.copy()
            pl_o[2] -= lo
            n = len(col)
            got = e.unary_batch(fr, sh, np.repeat(pl[None], n, 0), check=False)
            ref = o.unary_batch(fr, sh, np.repeat(pl_o[None], n, 0), check=False)
            pc.compare_maps(got, ref)
            checked += n
        del sub, o
    print(f"configs[4] size: {checked} cell evaluations of planes with |a| = 5/32 in slices 416-443 / 484-511 equal the oracle (tiled-copy taps, offsets > 2^31)")
    e.close()
    del vol
    torch.cuda.empty_cache()


def test_multi_gpu_rank_shape_whole_image_planes_vs_oracle(oracle_mod):
    """The per-rank workload of `bench.py --gpus N` (BASELINE configs[4]: 3000 x 2000 image, slices sharded over ranks): whole-image
    fronto-parallel and slanted planes through the march kernel (14 strips of 216 columns) against the oracle, full resolution."""
    import torch
    from localexpstereo_amd import api, synth
    H, W, D = 2000, 3000, 6
    guide = synth.make_guide(H, W, 1234)
    vol = synth.make_volume(D, H, W, 42)
    e = api.HipCostVolumeEnergy(guide, None, vol, None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1)
    o = pc.om.Oracle(guide, guide, vol, vol, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1)
    planes = np.array([[0, 0, 2.0, 0], [0.0007, -0.0005, 2.3, 0]], np.float32)
    full = [(0, 0, W, H)] * 2
    b = api.Batch(e, full, full, out_slabs=True)
    assert b.kernel_kind(0) == 1
    out = torch.empty((2, H, W), device="cuda", dtype=torch.float32)
    b.run(torch.from_numpy(planes).cuda().data_ptr(), out.data_ptr(), mode=0, check=True, planes_on_device=True)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for k in range(2):
        ref = o.unary_batch([(0, 0, W, H)], [(0, 0, W, H)], planes[k][None], check=True)
        pc.compare_maps(got[k], ref)
    e.close()


def test_bench_record_fields_on_one_gpu():
    """The driver's contract for the N = 1 line: one JSON line with the metric fields, the roofline object (with the co-bounds and the
    source of the traffic figure) and the CPU baseline with its H2 / H3 samples; the sub-records of the other workloads."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--sub-steps", "2", "--e2e", "0", "--cpu-planes", "8"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "co_bounds", "kernel_ms", "peak_achievable"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0.05 < rf["frac"] < 1.0
    if rf["traffic"] is not None:                    # quoted only when profiles/traffic.json was taken on exactly these kernel sources
        assert rf["traffic"] > rf["algorithmic_bytes_per_launch"] * 0.9 and rf["co_bounds"]["waves_per_simd"] == 3
    else:
        assert "not quoted" in rf["traffic_source"] or "unreadable" in rf["traffic_source"]
    for k in ("h2", "h3", "h3_batched", "n8_rank_shape"):
        assert d[k]["ms_per_step"] > 0, k
    assert d["h3_batched"]["ms_per_step"] < d["h3"]["ms_per_step"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["h2"]["value"] > 0 and cb["h3"]["value"] > 0
    assert cb["gpu_vs_oracle_max_abs_err_on_sample"] <= 2e-6


@pytest.mark.parametrize("world", [4, 8])
def test_bench_multi_rank_code_path_on_one_gpu(world):
    """`bench.py --gpus 4` / `--gpus 8` (the shape the first 8-GPU node will see: four ranks per view group) launched exactly as the driver does (torch.distributed.run, one process per rank), with all ranks on
    cuda:0 and gloo for the rendezvous / max-over-ranks reductions (LES_BENCH_BACKEND / LES_BENCH_ONE_DEVICE: test hooks, the
    measured configuration is RCCL with one GPU per rank): rank 0 prints one JSON line for the whole job."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LES_BENCH_BACKEND="gloo", LES_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29533 + world),
           os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--height", "500", "--width", "700", "--ndisp", "8", "--cpu-planes", "0"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["evals_per_step_per_gpu"] == 500 * 700 * 8
    # the leg WITH a collective (BASELINE configs[3]): view split x cell split -- four ranks = two per view group, so every disjoint set
    # ends with an all-gather inside its group, and the groups meet for the broadcast of the final label maps
    x = d["e2e_sharded"]
    assert "error" not in x, x
    assert x["seconds"] > 0 and len(x["bytes_exchanged_per_rank"]) == world and len(x["host_cut_seconds_per_rank"]) == world
    assert min(x["all_gathers_per_rank"]) > 0 and min(x["bytes_exchanged_per_rank"]) > 0
    assert x["bad_all_last"] is not None and x["bad_all_last"] < 50.0
