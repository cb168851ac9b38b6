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
        L.hbs_rmat_tail.restype = ctypes.c_void_p
        L.hbs_rmat_tail.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32,
                                    ctypes.c_uint32, ctypes.c_uint32]
        L.hbs_free.argtypes = [ctypes.c_void_p]
        for f in ("hbs_num_nodes", "hbs_num_edges", "hbs_raw_drawn"):
            getattr(L, f).restype = ctypes.c_uint64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        for f in ("hbs_ids", "hbs_row_ptr", "hbs_src"):
            getattr(L, f).restype = ctypes.c_void_p
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.hbs_stream_len.restype = ctypes.c_uint64
        L.hbs_stream_len.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.hbs_stream_lost_pairs.restype = ctypes.c_uint64
        L.hbs_stream_lost_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.hbs_stream_fill.restype = ctypes.c_uint64
        L.hbs_stream_fill.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.hbs_export_edges.restype = ctypes.c_uint64
        L.hbs_export_edges.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]
        _lib = L
    return _lib


class RmatGraph:
    """n touched hosts, m unique non-self edges; ids ascending; CSR by destination."""

    def __init__(self, scale, m_target, seed=SEED, threads=0, tail=None):
        """tail = (tail_permille, ratio_permille, fanin): the long-tail variant (hb_synth.cpp: a levelled DAG of
        tail_permille/1000 * n_core extra hosts hanging off the core, T grows by the tail depth)."""
        L = _load()
        if tail:
            self._g = L.hbs_rmat_tail(scale, m_target, seed, threads, int(tail[0]), int(tail[1]), int(tail[2]))
        else:
            self._g = L.hbs_rmat(scale, m_target, seed, threads)
        if not self._g:
            raise ValueError("hbs_rmat failed (scale must be 1..31; 1..30 with a tail)")
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

    def stream_len(self, salt=0):
        """Records of the streamed export (salt 0: the clean edges permuted; salt 2: plus flagged-first extras, their
        clean copies and flagged duplicates - a stream that the reference semantics reduce to exactly this graph)."""
        return int(_load().hbs_stream_len(self._g, salt))

    def stream_lost_pairs(self, salt=0):
        return int(_load().hbs_stream_lost_pairs(self._g, salt))

    def stream_fill(self, out, first, salt=0):
        """Fills out[:k] (EDGE records, any writable contiguous buffer) with records first.. of the stream; returns k."""
        assert out.dtype == EDGE and out.flags["C_CONTIGUOUS"]
        return int(_load().hbs_stream_fill(self._g, salt, first, len(out), out.ctypes.data))

    def stream(self, salt=0, slab=1 << 22, buf=None):
        """Yields the stream slab by slab (views of one reused buffer)."""
        total = self.stream_len(salt)
        if buf is None:
            buf = np.zeros(min(slab, max(total, 1)), dtype=EDGE)
        at = 0
        while at < total:
            k = self.stream_fill(buf, at, salt)
            yield buf[:k]
            at += k

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
    # long convergence tail (real host graphs take tens of passes; plain R-MAT converges in 9-10): the C3 core plus
    # a levelled DAG of 30 % extra hosts, level sizes shrinking by 0.8, 10 in-links per tail host
    "LT": dict(scale=24, m=200_000_000, tail=(300, 800, 10), label="long-tail: C3 core + 30% levelled-DAG tail (r=0.8, fan-in 10)"),
    "LT2": dict(scale=21, m=20_000_000, tail=(300, 800, 10), label="long-tail: C2 core + 30% levelled-DAG tail (r=0.8, fan-in 10)"),
}


class CachedGraph:
    """The reduced form of a generated graph (ids, row_ptr, src) read back from .npy files: for profiling scripts that run
    the same big config several times in one session (HB_SYNTH_CACHE=<dir>); no record-stream export."""

    def __init__(self, prefix):
        self.ids = np.load(prefix + "_ids.npy", mmap_mode="r").view(U128).reshape(-1)
        self.row_ptr = np.load(prefix + "_row_ptr.npy", mmap_mode="r")
        self.src = np.load(prefix + "_src.npy", mmap_mode="r")
        self.n, self.m = len(self.ids), len(self.src)

    @staticmethod
    def save(g, prefix):
        np.save(prefix + "_ids.npy", np.ascontiguousarray(g.ids).view(np.uint64).reshape(-1, 2))
        np.save(prefix + "_row_ptr.npy", g.row_ptr)
        np.save(prefix + "_src.npy", g.src)

    def id_low64(self):
        return np.ascontiguousarray(self.ids["lo"])

    def close(self):
        pass


def make_config(name):
    """(graph, scale, label) of a named BASELINE config, or of 'scale:m' / 'scale:m:tail_permille:ratio_permille:fanin'."""
    if name in CONFIGS:
        cfg = CONFIGS[name]
        label = "%s (R-MAT scale %d, a,b,c,d=.57,.19,.19,.05, seed 0x5712AC7)" % (cfg["label"], cfg["scale"])
        cache = os.environ.get("HB_SYNTH_CACHE")
        if cache:
            prefix = os.path.join(cache, "hb_synth_%s" % name)
            if os.path.exists(prefix + "_src.npy"):
                return CachedGraph(prefix), cfg["scale"], label
        g = RmatGraph(cfg["scale"], cfg["m"], tail=cfg.get("tail"))
        if cache:
            os.makedirs(cache, exist_ok=True)
            CachedGraph.save(g, prefix)
        label = "%s (R-MAT scale %d, a,b,c,d=.57,.19,.19,.05, seed 0x5712AC7)" % (cfg["label"], cfg["scale"])
        return g, cfg["scale"], label
    parts = [int(x) for x in name.split(":")]
    scale, m_target = parts[0], parts[1]
    tail = tuple(parts[2:5]) if len(parts) >= 5 else None
    label = "R-MAT scale %d / %d edges%s" % (scale, m_target, " + tail %s" % (tail,) if tail else "")
    return RmatGraph(scale, m_target, tail=tail), scale, label
