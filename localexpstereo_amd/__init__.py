"""localexpstereo_amd -- MI355X-native matching-cost path of LocalExpStereo.

The package holds only what the hot path needs: csrc/ (HIP kernels + the C-ABI shared library),
host/ (the C++ host-side mirror of the reference's StereoEnergy operator interface), a ctypes
binding of the C-ABI (api.py) and seeded synthetic inputs (synth.py).
"""
import os as _os

# The host cuts of the graph-cut iterations run in OpenMP teams that are idle while the GPU works; with the default ("active") wait policy
# their threads spin through those gaps, and the two teams of a two-view run then spin against each other (measured: 6.35 s instead of 5.9 s
# at the Adirondack shape).  The policy is read when the OpenMP runtime starts, so it is set here, at package import, unless the caller chose one.
_os.environ.setdefault("OMP_WAIT_POLICY", "passive")

__version__ = "0.1.0"
