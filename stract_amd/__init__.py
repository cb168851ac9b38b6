"""stract_amd - MI355X-native HyperBall harmonic centrality for Stract's webgraph.

Only what the hot path needs: the C-ABI library (csrc/, include/hyperball.h), its ctypes
binding (_lib), the mirror of the reference's `HarmonicCentrality` interface (harmonic),
synthetic graph input (synth) and the multi-GPU driver (dist).
"""
from ._lib import Context, HyperballError, device_count  # noqa: F401
from .harmonic import EdgeListGraph, HarmonicCentrality  # noqa: F401
