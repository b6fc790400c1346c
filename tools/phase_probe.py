"""Share of a workgroup's time spent in each phase of the strip kernel (G, H1, V, H2, F), measured with clock64() around the
barriers in a -DLES_PHASE_TIMING build of the same sources.  Build the instrumented library first (here, or in the build
container -- it travels with the snapshot):  python tools/phase_probe.py --build ; then on the GPU: python tools/phase_probe.py"""
import ctypes as C, os, sys, json, subprocess
sys.path.insert(0, os.getcwd())
LIB = os.path.join(os.getcwd(), "localexpstereo_amd/csrc/libles_phase_timing.so")
if "--build" in sys.argv:
    from localexpstereo_amd import build
    subprocess.check_call([build._hipcc()] + build.HIPCC_FLAGS + ["-DLES_MARCH_LAB", "-DLES_PHASE_TIMING", os.path.join(build.CSRC, "les_hip.hip"), "-o", LIB], cwd=build.CSRC)
    print("built", LIB)
    sys.exit(0)
os.environ["LES_HIP_LIB"] = os.environ.get("PHASE_LIB", LIB)
import torch, numpy as np
from localexpstereo_amd import api, synth
H, W, D = 1000, 1500, 256
dev = torch.device("cuda", 0)
guide = synth.make_guide(H, W, 1234)
vol = torch.rand((D, H, W), device=dev, dtype=torch.float32)
e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1, volumes_on_device=True, shape=(D, H, W))
slanted = "--slanted" in sys.argv
planes = torch.from_numpy(synth.slanted_planes(D, H, W, D - 1, seed=7) if slanted else synth.fronto_planes(D)).to(dev)
out = torch.empty((D, H, W), device=dev, dtype=torch.float32)
full = [(0, 0, W, H)] * D
b = api.Batch(e, full, full, out_slabs=True)
L = e.L
buf = (C.c_ulonglong * 12)()
b.run(planes.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
L.les_hip_debug_phases(buf)
for it in range(2):
    b.run(planes.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
    L.les_hip_debug_phases(buf)
    kind = b.kernel_kind(0)
    if kind == 1:
        v = [buf[i] for i in range(12)]
        for r, name in enumerate("ACD"):
            n = max(1, v[6 + r])
            print(f"march role {name}: wave-ticks {v[6 + r]}, cycles per tick: compute {v[2 * r] / n:.0f} (row loop {v[9 + r] / n:.0f}), barrier wait {v[2 * r + 1] / n:.0f}")
        continue
    v = [buf[i] for i in range(6)]
    tot = sum(v[:5])
    nblk = (H + 40 + 6) // 7 if kind == 1 else 49.5
    print("kernel", "march (phases A/B1/C/B2/D)" if kind == 1 else "strip (phases G/H1/V/H2/F)", "WGs", v[5], "cycles per WG", tot / v[5], "phase shares:", [round(100.0 * x / tot, 1) for x in v[:5]],
          "per block-phase cycles", [round(x / v[5] / nblk) for x in v[:5]])
