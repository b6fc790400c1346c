"""RANSAC proposer: device vs oracle agreement (planes bit for bit, RNG states) over several layers and seeds."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import parity_cases as pc
from localexpstereo_amd import api
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "hip" else None
pr = pc.synth_pair(lib, 200, 260, 32) if lib is None else pc.synth_pair(lib, 90, 120, 16)
H, W, D = pr.H, pr.W, pr.D
tot = same_bits = same_state = close = 0
for unit in ((8, 15, 40) if lib is None else (8, 15)):
    layer = pc.om.Layer(W, H, 20, unit)
    for si in (0, 5, len(layer.sets) - 1):
        cells = layer.sets[si]
        units = layer.unit[cells]
        n = len(cells)
        b = api.Batch(pr.e, layer.filter[cells], layer.shared[cells]); b.set_units(units)
        for seed, noise in ((3, 0.3), (4, 0.02), (5, 2.0)):
            labels = pc._label_map(H, W, D, seed, noise=noise)
            d_lab, d_rng, d_pl = api.DeviceBuffer(pr.e, labels.nbytes), api.DeviceBuffer(pr.e, 8 * n), api.DeviceBuffer(pr.e, 16 * n)
            seeds = pc._seeds(n, seed + 11 * unit)
            d_lab.upload(labels); d_rng.upload(seeds)
            b.propose(api.PROPOSE_RANSAC, d_lab.ptr, d_rng.ptr, d_pl.ptr, m=0)
            pr.e.synchronize()
            got = d_pl.download((n,), api.PLANE_DT); st = d_rng.download((n,), np.uint64)
            ref, rst = pc._oracle_proposals(api.PROPOSE_RANSAC, labels, W, units, seeds, 0, 0.0, float(D - 1))
            g4, r4 = got.view(np.float32).reshape(n, 4), ref.view(np.float32).reshape(n, 4)
            eq = np.all(g4.view(np.uint32) == r4.view(np.uint32), axis=1)
            cl = np.all(np.abs(g4 - r4) <= 1e-4 * np.maximum(1, np.abs(r4)), axis=1)
            tot += n; same_bits += int(eq.sum()); same_state += int((st == rst).sum()); close += int(cl.sum())
            bad = np.where(~cl)[0][:2]
            for i in bad:
                print("  differs: unit", unit, "set", si, "seed", seed, "cell", i, "dev", g4[i], "ref", r4[i], "state eq", st[i] == rst[i])
            for d in (d_lab, d_rng, d_pl): d.free()
        b.destroy()
print(f"RANSAC proposals: {tot} cells, bit-identical planes {same_bits} ({same_bits / tot:.4f}), within 1e-4 {close} ({close / tot:.4f}), identical RNG states {same_state} ({same_state / tot:.4f})")
