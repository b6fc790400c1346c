"""localexpstereo_amd -- MI355X-native matching-cost path of LocalExpStereo.

The package holds only what the hot path needs: csrc/ (HIP kernels + the C-ABI shared library),
host/ (the C++ host-side mirror of the reference's StereoEnergy operator interface), a ctypes
binding of the C-ABI (api.py) and seeded synthetic inputs (synth.py).
"""
__version__ = "0.1.0"
