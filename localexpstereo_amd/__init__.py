"""localexpstereo_amd -- MI355X-native matching-cost path of LocalExpStereo.

The package holds only what the hot path needs: csrc/ (HIP kernels + the C-ABI shared library),
host/ (the C++ host-side mirror of the reference's StereoEnergy operator interface), a ctypes
binding of the C-ABI (api.py) and seeded synthetic inputs (synth.py).
"""
# (The host cuts of the graph-cut iterations run in OpenMP teams that are idle while the GPU works.  With the default "active" wait policy their
# threads spin through those gaps: 6.35 s instead of 5.9 s for a two-view run at the Adirondack shape.  The policy is read once, when the OpenMP
# runtime starts, and it is process-wide, so a library import must not set it: the entry points do -- bench.py, tools/e2e_bench.py, tools/run_demo.py
# put OMP_WAIT_POLICY=passive into the environment before anything starts OpenMP -- and an embedding application chooses for itself.)
__version__ = "0.1.0"
