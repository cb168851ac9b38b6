"""Synthetic webgraphs for benches and tests (R-MAT): ctypes wrapper of
stract_amd/lib/libhb_synth.so (host-only C++, stract_amd/csrc/hb_synth.cpp)."""
import ctypes
import os

import numpy as np

from ._lib import EDGE, U128

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhb_synth.so")
SEED = 0x5712AC7
_lib = None


def _load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        L.hbs_rmat.restype = ctypes.c_void_p
        L.hbs_rmat.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        L.hbs_free.argtypes = [ctypes.c_void_p]
        for f in ("hbs_num_nodes", "hbs_num_edges", "hbs_raw_drawn"):
            getattr(L, f).restype = ctypes.c_uint64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        for f in ("hbs_ids", "hbs_row_ptr", "hbs_src"):
            getattr(L, f).restype = ctypes.c_void_p
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.hbs_export_edges.restype = ctypes.c_uint64
        L.hbs_export_edges.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]
        _lib = L
    return _lib


class RmatGraph:
    """n touched hosts, m unique non-self edges; ids ascending; CSR by destination."""

    def __init__(self, scale, m_target, seed=SEED, threads=0):
        L = _load()
        self._g = L.hbs_rmat(scale, m_target, seed, threads)
        if not self._g:
            raise ValueError("hbs_rmat failed (scale must be 1..31)")
        self.scale = scale
        self.n = L.hbs_num_nodes(self._g)
        self.m = L.hbs_num_edges(self._g)
        self.raw_drawn = L.hbs_raw_drawn(self._g)

        def view(ptr, ctype, shape):
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape)

        self.ids = view(L.hbs_ids(self._g), ctypes.c_uint64, (self.n, 2)).view(U128).reshape(self.n) if self.n else np.zeros(0, U128)
        self.row_ptr = view(L.hbs_row_ptr(self._g), ctypes.c_uint64, (self.n + 1,))
        self.src = view(L.hbs_src(self._g), ctypes.c_uint32, (self.m,)) if self.m else np.zeros(0, np.uint32)

    def id_low64(self):
        return np.ascontiguousarray(self.ids["lo"])

    def edges(self, salt=0, salt_seed=1):
        """Raw SmallEdge records (stream order); salt=1 adds flagged / duplicate / self-loop
        records that exercise the reference's ingest semantics."""
        L = _load()
        cap = L.hbs_export_edges(self._g, None, 0, salt, salt_seed)
        out = np.zeros(cap, dtype=EDGE)
        k = L.hbs_export_edges(self._g, out.ctypes.data, cap, salt, salt_seed)
        assert k == cap
        return out

    def close(self):
        if self._g:
            _load().hbs_free(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# BASELINE.json configs: id-space scale and unique-edge target chosen so that the
# TOUCHED host count lands at the config's host count (SURVEY.md §8 preamble).
CONFIGS = {
    "C1": dict(scale=14, m=100_000, label="10k-host / 100k-edge"),
    "C2": dict(scale=21, m=20_000_000, label="1M-host / 20M-edge"),
    "C3": dict(scale=24, m=200_000_000, label="10M-host / 200M-edge"),
    "C4": dict(scale=28, m=2_000_000_000, label="100M-host / 2B-edge"),
    "C5": dict(scale=30, m=5_000_000_000, label="~300M-host / ~5B-edge (CommonCrawl-scale host graph, synthetic)"),
}
