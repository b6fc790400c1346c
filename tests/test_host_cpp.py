"""C++ host side above the C ABI (localexpstereo_amd/host/): LayerManager geometry against the oracle
(no GPU needed) and the les_host_demo end-to-end run (drop-in operator from OpenMP threads + device-
resident PatchMatch iterations) on the MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest


@pytest.fixture(scope="module")
def demo():
    import os
    from localexpstereo_amd import build
    exe = os.path.join(build.HOST, "les_host_demo")
    if os.path.exists(exe) and os.path.exists(build.HIP_SO):
        return exe                     # prebuilt by __graft_entry__.build() (the GPU box only uses the prebuilt files)
    build.build_hip()
    exe = build.build_host()
    assert exe
    return exe


@pytest.mark.parametrize("W,H,units", [(450, 375, (5, 15, 25)), (1436, 992, (14, 43, 129)), (1500, 1000, (15, 45, 135)),
                                       (3000, 2000, (30, 90, 270)), (120, 96, (14, 7)), (97, 61, (10, 13))])
def test_host_layer_manager_matches_oracle(demo, oracle_mod, W, H, units):
    for u in units:
        out = subprocess.run([demo, "layers", str(W), str(H), "20", str(u)], capture_output=True, text=True, check=True).stdout.split("\n")
        wb, hb, nsets = (int(v) for v in out[0].split())
        L = oracle_mod.Layer(W, H, 20, u)
        assert (wb, hb, nsets) == (L.width_blocks, L.height_blocks, len(L.sets))
        n = wb * hb
        rects = np.array([[int(v) for v in line.split()] for line in out[1:1 + n]], np.int32)
        for k, ref in enumerate((L.unit, L.shared, L.filter)):
            got = rects[:, 4 * k:4 * k + 4]
            exp = np.stack([ref["x"], ref["y"], ref["w"], ref["h"]], 1)
            assert np.array_equal(got, exp), f"unit {u} rect kind {k}"
        for s, line in zip(L.sets, out[1 + n:1 + n + nsets]):
            assert [int(v) for v in line.split()] == s.tolist()


@pytest.mark.gpu
def test_host_demo_runs_on_gpu(demo):
    r = subprocess.run([demo, "run", "240", "160", "32", "2"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "les_host_demo: OK" in r.stdout


@pytest.mark.gpu
def test_host_demo_on_the_python_drivers_scene(demo, tmp_path):
    """The C++ driver on the scene of tools/e2e_bench.py (written to raw files by tools/dump_scene.py, left volume as the device ingest
    leaves it): same data, parameters and layers as the Python driver's end-to-end run -- and the same wall-clock (2.0 s on the MI355X box)."""
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "scene")
    subprocess.run([sys.executable, os.path.join(root, "tools", "dump_scene.py"), "--out", d], check=True, timeout=600, cwd=root)
    env = dict(os.environ, OMP_WAIT_POLICY="passive")
    r = subprocess.run([demo, "scene", d, "5", "2"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "les_host_demo: OK" in r.stdout
    sec = float(re.search(r"optimiser ([0-9.]+) s", r.stdout).group(1))
    bad = float(re.search(r"bad1.0=([0-9.]+)%", r.stdout).group(1))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "host_demo_scene.log"), "w") as f:
        f.write(r.stdout)
    assert bad < 15.0, bad           # (the Python driver ends at 11.1 % on this scene; occlusions of the synthetic pair)
    assert sec < 2.4, sec            # (1.6-1.8 s on the MI355X boxes with every cut on the GPU: 1.3 x)


@pytest.mark.gpu
def test_host_demo_full_size_on_gpu(demo):
    """The C++ host (PMStereo::runDevice) at the Adirondack-H shape: the MidV3 loop with device-built graphs and device cuts; its
    wall-clock stands next to the Python driver's (tools/e2e_bench.py) in DESIGN.md."""
    import os
    env = dict(os.environ, OMP_WAIT_POLICY="passive")
    r = subprocess.run([demo, "full", "1436", "992", "256", "5", "2"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "les_host_demo: OK" in r.stdout
    import re
    sec = float(re.search(r"optimiser ([0-9.]+) s", r.stdout).group(1))
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "host_demo_full.log"), "w") as f:
        f.write(r.stdout)
    # round 5: 3.75-3.96 s on the MI355X boxes with every cut on the GPU (the coarse layers by the tiled solver; 11.5 s with their cuts on the host cores as in
    # rounds 2-4: `les_host_demo full 1436 992 256 5 2 0`).  The bound is 1.3 x the largest measured.
    assert "host cuts 0.000 s" in r.stdout, "a lock-step was cut on the host"
    assert sec < (5.2 if os.environ.get("LES_TEST_STRICT_TIMING") else 10.0), sec      # (default: north_star's 10 s; LES_TEST_STRICT_TIMING=1: 1.3 x the measured)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_host_demo_sharded_ranks_on_one_gpu(demo, world):
    """The C++ host's multi-rank path (PMStereo::runDevice with rank / world: bands of cells per rank, per-set tile exchange through
    les_hip_exchange_pack / _unpack) with 2 and 3 ranks as host threads on the one GPU and a loop-back transport: PatchMatch + graph-cut
    iteration, every rank bit-equal to the single-rank run."""
    r = subprocess.run([demo, "ranks", "240", "160", "32", str(world)], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "les_host_demo: OK" in r.stdout


def test_push_relabel_equals_bk_on_real_graphs(monkeypatch):
    """Two 129 x 129 crops of a coarse-layer lock-step dumped from the Adirondack-shape run (89 % and 20 % of the nodes switch: the hard
    kind): the push-relabel solver and the Boykov-Kolmogorov solver of liblocalexp_host.so must return the same mask and the same flow,
    and the mask must be a minimum cut according to networkx."""
    import importlib
    import networkx as nx
    from localexpstereo_amd import api, build
    build.build_host_lib()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hard_cells.npz"))
    res = {}
    for solver, thr in (("bk", "0"), ("pr", "1"), ("plain", "0")):
        # the threshold is read once per process: run each solver in its own interpreter
        code = ("import numpy as np, sys; from localexpstereo_amd import gc as lgc, api\n"
                "z = np.load(sys.argv[1]); out = {}\n"
                "for k in z.files:\n"
                "    p = np.ascontiguousarray(z[k].reshape(-1), np.float32); h, w = z[k].shape[:2]\n"
                "    m = np.zeros(w * h, np.uint8); f = np.zeros(1)\n"
                "    lgc.solve_prebuilt(api._rects(np.array([(0, 0, w, h)], np.int32)), p, np.array([0], np.int64), m, nthreads=1, flows_out=f)\n"
                "    out[k + '_mask'] = m; out[k + '_flow'] = f\n"
                "np.savez(sys.argv[2], **out)\n")
        outp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"hard_cells_{solver}.npz")
        env = dict(os.environ, LES_GC_PUSH_RELABEL_MIN_NODES=thr, LES_GC_BK_OPS_PER_NODE="1")     # ("pr": the budget of the BK phase runs out at once)
        if solver == "plain":
            env["LES_GC_PREPUSH"] = "0"           # the graphs loaded as they come: both search trees, segments read off the trees
        subprocess.run([sys.executable, "-c", code, os.path.join(os.path.dirname(__file__), "golden", "hard_cells.npz"), outp], check=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        res[solver] = np.load(outp)
    for k in z.files:
        mb, mp = res["bk"][k + "_mask"], res["pr"][k + "_mask"]
        assert np.array_equal(mb != 0, mp != 0), f"{k}: {int(((mb != 0) != (mp != 0)).sum())} nodes differ between the two solvers"
        fb, fp = float(res["bk"][k + "_flow"][0]), float(res["pr"][k + "_flow"][0])
        assert abs(fb - fp) <= 1e-6 * abs(fb), (fb, fp)
        assert 0 < (mp != 0).mean() < 1
        # the default path (local pre-push while loading, search from the source side, segments by residual reachability) against the plain one
        ml = res["plain"][k + "_mask"]
        assert np.array_equal(mb != 0, ml != 0), f"{k}: {int(((mb != 0) != (ml != 0)).sum())} nodes differ between the pre-pushed and the plain search"
        assert abs(fb - float(res["plain"][k + "_flow"][0])) <= 1e-6 * abs(fb)
    # independent check of one of them: networkx max-flow value == flow, and the mask is a cut of that capacity
    from tests import parity_cases as pc
    k = "cell5"
    h, w = z[k].shape[:2]
    ref_flow, ref_src = pc._grid_graph_reference(z[k].reshape(-1, 5), w, h)
    assert abs(ref_flow - float(res["pr"][k + "_flow"][0])) <= 1e-5 * ref_flow
    cap = pc._cut_capacity(z[k].reshape(-1, 5), w, h, res["pr"][k + "_mask"] != 0)
    assert abs(cap - ref_flow) <= 1e-5 * ref_flow
    assert int(((res["pr"][k + "_mask"] != 0) != ref_src).sum()) <= 2            # (float ties only)


def test_host_graph_cut_selfcheck(demo):
    """Local expansion moves on the host (ExpansionMove.h over MaxFlow.h): brute-force optimality on tiny regions, the
    reference's flow == energy self-check (LES/FastGCStereo.h:561-594) on every move, monotone energy, convergence."""
    from localexpstereo_amd import build
    exe = build.build_host_selfcheck()
    r = subprocess.run([exe, "120", "80", "24", "2"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "gc_selfcheck: OK" in r.stdout


def _maxflow_case(demo, tmp_path, n, tw, edges):
    path = tmp_path / "g.txt"
    with open(path, "w") as f:
        f.write(f"{n} {len(edges)}\n")
        for a, b in tw:
            f.write(f"{a:.9g} {b:.9g}\n")
        for i, j, c, r in edges:
            f.write(f"{i} {j} {c:.9g} {r:.9g}\n")
    out = subprocess.run([demo, "maxflow", str(path)], capture_output=True, text=True, check=True).stdout.split()
    return float(out[0]), [int(ch) for ch in out[1]]


def test_host_maxflow_matches_networkx(demo, tmp_path):
    """N2: the host min-cut (host/MaxFlow.h) against networkx on random grid-like graphs with float capacities:
    same flow value, the reported segments form a cut of that value, and the sink side is the set of nodes that can
    still reach the sink (so ties go to SOURCE like the library the reference links, LES/FastGCStereo.h:557)."""
    import networkx as nx
    rng = np.random.default_rng(0)
    for trial in range(12):
        h, w = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        n = h * w
        tw = np.round(rng.uniform(0, 4, (n, 2)), 3).astype(np.float32)
        if trial % 3 == 0:
            tw[rng.random(n) < 0.3] = 0            # nodes attached to no terminal
        edges = []
        for y in range(h):
            for x in range(w):
                for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
                    yy, xx = y + dy, x + dx
                    if 0 <= yy < h and 0 <= xx < w:
                        c = float(np.float32(round(rng.uniform(0, 2), 3))) if rng.random() > 0.15 else 0.0
                        edges.append((y * w + x, yy * w + xx, c, 0.0 if trial % 2 else c))
        flow, seg = _maxflow_case(demo, tmp_path, n, tw, edges)
        G = nx.DiGraph()
        G.add_nodes_from(["s", "t"])
        base = 0.0
        for i, (a, b) in enumerate(tw):
            m = min(a, b)
            base += m
            if a - m > 0:
                G.add_edge("s", i, capacity=float(a - m))
            if b - m > 0:
                G.add_edge(i, "t", capacity=float(b - m))
        for i, j, c, r in edges:
            for u, v, cap in ((i, j, c), (j, i, r)):
                if cap > 0:
                    G.add_edge(u, v, capacity=G[u][v]["capacity"] + cap if G.has_edge(u, v) else cap)
        ref, (S, T) = nx.minimum_cut(G, "s", "t")
        assert abs(flow - (ref + base)) <= 1e-4 * max(1.0, ref + base)
        # capacity of the cut our segments define == the flow
        cut = base
        for i, (a, b) in enumerate(tw):
            m = min(a, b)
            cut += (b - m) if seg[i] == 0 else (a - m)
        for i, j, c, r in edges:
            if seg[i] == 0 and seg[j] == 1:
                cut += c
            if seg[j] == 0 and seg[i] == 1:
                cut += r
        assert abs(cut - flow) <= 1e-4 * max(1.0, flow)
        # sink side == nodes that can reach t in the residual graph of a maximum flow (unique minimal sink set)
        R = nx.algorithms.flow.preflow_push(G, "s", "t")
        reach = {"t"}
        stack = ["t"]
        while stack:
            v = stack.pop()
            for u in R.predecessors(v):
                if u not in reach and R[u][v]["capacity"] - R[u][v]["flow"] > 1e-9:
                    reach.add(u)
                    stack.append(u)
        for i in range(n):
            assert (seg[i] == 1) == (i in reach), (trial, i)


def test_host_banded_solver_matches_plain():
    """A region of more than 40 000 nodes goes through the band-parallel max-flow (and parallel node load / read-out) in
    les_gc_solve_prebuilt; the same move through les_gc_expansion_moves uses the plain search.  Same labels."""
    from localexpstereo_amd import build, gc, synth
    build.build_host_lib()
    H, W = 230, 260
    im = synth.make_guide(H, W, 3)
    rng = np.random.default_rng(4)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    lab = np.zeros((H, W, 4), np.float32)
    lab[..., 2] = 6 + 3 * ((xs // 40 + ys // 35) % 3)
    lab[..., 0] = 0.01
    cur = (0.25 + 0.15 * np.sin(0.05 * xs) + rng.uniform(0, 0.03, (H, W))).astype(np.float32)
    prop = (0.25 + 0.15 * np.cos(0.04 * ys) + rng.uniform(0, 0.03, (H, W))).astype(np.float32)
    region = [(5, 4, 240, 215)]                                   # 51 600 nodes
    plane = [(0.005, -0.004, 7.5, 0.0)]
    g = gc.GraphCut(im, None, lambda_=0.3)
    g.labels[0][...] = lab
    g.costs[0][...] = cur
    off = np.zeros(1, np.int64)
    payload, flow0 = g.build_graphs(region, plane, prop, off)
    masks = np.zeros(240 * 215, np.uint8)
    gc.solve_prebuilt(region, payload, off, masks)               # banded: one cell, spare threads
    gap = g.expansion_moves(region, plane, prop, check=True)     # plain search on the host-built graph
    assert gap <= 1e-5
    took = (g.labels[0][4:219, 5:245].view(np.uint32) != lab[4:219, 5:245].view(np.uint32)).any(axis=2)
    assert np.array_equal(took, masks.reshape(215, 240) > 0)
    assert 0.05 < took.mean() < 0.95                              # a real cut, crossing every band
    g.close()
