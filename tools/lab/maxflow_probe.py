"""Iteration counts, relabel sweeps and kernel time of the device max-flow (csrc/les_maxflow.h) on layer-0 lock-steps of a synthetic
Adirondack-shape scene, next to the host team on the same graphs.  Needs a measurement build that reports the counts through the
status / flows outputs:  bash tools/build_variant.sh mfdbg -DLES_MF_DEBUG_ITERS   (MF_LIB=<other variant>.so selects another one,
e.g. built with -DLES_MF_G=<period of the global relabelling>)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["LES_HIP_LIB"] = os.path.join(os.getcwd(), "localexpstereo_amd/csrc", os.environ.get("MF_LIB", "libles_mfdbg.so"))
import numpy as np, torch
from localexpstereo_amd import api, pm, gc as lgc, synth
from localexpstereo_amd.synth import make_scene, ad_volume
H, W, D = 992, 1436, 256
imL, imR, gt = make_scene(H, W, D)
volL = ad_volume(imL, imR, D, "cuda")
e = api.HipCostVolumeEnergy(imL, imR, volL.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1, volumes_on_device=True, shape=(D, H, W))
table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 7)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
r = pm.PMRunner(e, (14, 43, 129), table, seed=1, device="cuda")
r.init_labels()
for it in range(2):
    r.iteration(it)
g = lgc.GraphCut(imL, imR, lambda_=0.5 * 20 / 1.0 if False else 10.0, th_smooth=1.0, omega=10.0, epsilon=0.01)
r.begin_gc(g)
# one lock-step by hand: set 3 of layer 0, proposal kind RANDOM
sh = r.shards[0][3]
r._gc_buffers(sh)
p = g.params
allit = []
for kind in (api.PROPOSE_EXPANSION, api.PROPOSE_EXPANSION, api.PROPOSE_RANDOM, api.PROPOSE_RANSAC):
    sh.batch.propose(kind, r.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr(), m=0)
    sh.batch.run(sh.planes.data_ptr(), r.prop.data_ptr(), mode=0, check=True, planes_on_device=True)
    sh.batch.expansion_graph(sh.planes.data_ptr(), r.labels.data_ptr(), r.cur.data_ptr(), r.prop.data_ptr(), sh.payload.data_ptr(), mode=0,
                             lambda_=p["lambda_"], th_smooth=p["th_smooth"], omega=p["omega"], epsilon=p["epsilon"])
    st = torch.zeros(sh.n, dtype=torch.int32, device="cuda")
    sw = torch.zeros(sh.n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t = time.perf_counter()
    sh.batch.solve_graphs(sh.payload.data_ptr(), sh.masks.data_ptr(), st.data_ptr(), sw.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    its = -st.cpu().numpy()
    ph = sh.payload.cpu().numpy(); mh = np.zeros(sh.graph_nodes, np.uint8)
    t = time.perf_counter(); lgc.solve_prebuilt(sh.regions, ph, sh.graph_off, mh); th = time.perf_counter() - t
    dm = sh.masks.cpu().numpy()[: sh.graph_nodes]
    print(f"kind {kind}: {sh.n} cells, device {dt * 1e3:.2f} ms, host {th * 1e3:.2f} ms; iterations median {np.median(its):.0f} mean {its.mean():.1f} max {its.max()}, relabel sweeps median {np.median(sw.cpu().numpy()):.0f} max {sw.max().item():.0f}; "
          f"nodes that differ from the host cut: {int(((dm != 0) != (mh != 0)).sum())} of {sh.graph_nodes}; changed {100 * (mh != 0).mean():.2f} %")
