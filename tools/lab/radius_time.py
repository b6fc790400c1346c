#!/usr/bin/env python
"""H1-shaped pass (1500 x 1000 x 64 fronto-parallel planes) at small guided-filter radii: the march kernel (radii 1 .. 3 have instantiations since
round 6) against the strip kernel (LES_HIP_KERNEL=strip), milliseconds per pass.   python tools/lab/radius_time.py   (GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    import torch
    from localexpstereo_amd import api, synth
    windR = int(sys.argv[1])
    H, W, D = 1000, 1500, 64
    dev = torch.device("cuda:0")
    guide = synth.make_guide(H, W, 1234)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    vol = torch.rand((D, H, W), device=dev, dtype=torch.float32, generator=gen)
    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=windR, eps=1e-3, th_col=0.5, max_disp=D - 1, device=0, volumes_on_device=True, shape=(D, H, W))
    e.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    out = torch.empty((D, H, W), device=dev, dtype=torch.float32)
    planes = torch.from_numpy(synth.fronto_planes(D)).to(dev)
    full = [(0, 0, W, H)] * D
    b = api.Batch(e, full, full, out_slabs=True)
    for _ in range(3):
        b.run(planes.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        b.run(planes.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
    t1.record(); torch.cuda.synchronize()
    print(f"windR {windR} (radius {windR // 2}) kernel kind {b.kernel_kind(0)} ({'march' if b.kernel_kind(0) else 'strip'}): {t0.elapsed_time(t1) / 20:.3f} ms per pass of {D} planes")
else:
    for windR in (2, 4, 6, 8):
        for strip in (0, 1):
            env = dict(os.environ, LES_HIP_QUIET="1")
            if strip:
                env["LES_HIP_KERNEL"] = "strip"
            subprocess.run([sys.executable, os.path.abspath(__file__), str(windR)], env=env, check=False)
