"""ctypes wrapper of the CPU oracle (oracle/libhb_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py - never from stract_amd/ (the product)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhb_oracle.so")
FRONTIER, LITERAL = 1, 2
BSEARCH_RUST_1_82, BSEARCH_CLASSIC = 0, 1
_lib = None

U128 = np.dtype([("lo", "<u8"), ("hi", "<u8")])
EDGE = np.dtype([("from", U128), ("to", U128), ("rel_flags", "<u8")])


class PassStats(ctypes.Structure):
    _fields_ = [("pass_", ctypes.c_uint64), ("active_edges", ctypes.c_uint64), ("touched", ctypes.c_uint64),
                ("changed", ctypes.c_uint64), ("has_changes", ctypes.c_int)]


class _U128S(ctypes.Structure):
    _fields_ = [("lo", ctypes.c_uint64), ("hi", ctypes.c_uint64)]


class FaithfulStats(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("m_unique", ctypes.c_uint64), ("m_eff", ctypes.c_uint64),
                ("passes", ctypes.c_uint64), ("passes_exact", ctypes.c_uint64), ("seconds_loop", ctypes.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = ctypes.CDLL(LIB_PATH)
    P, U64 = ctypes.c_void_p, ctypes.c_uint64
    L.hbo_hll_add.argtypes = [P, U64]
    L.hbo_hll_merge.argtypes = [P, P]
    L.hbo_hll_size.restype = U64
    L.hbo_hll_size.argtypes = [P]
    L.hbo_hll_size_ex.restype = U64
    L.hbo_hll_size_ex.argtypes = [P, ctypes.c_int, P, P]
    L.hbo_hll_estimate_bias.restype = ctypes.c_double
    L.hbo_hll_estimate_bias.argtypes = [ctypes.c_double, ctypes.c_int]
    L.hbo_hll_bias_first_index.restype = ctypes.c_int
    L.hbo_hll_bias_first_index.argtypes = [ctypes.c_double, ctypes.c_int]
    L.hbo_kahan_add.argtypes = [P, P, ctypes.c_double]
    L.hbo_dense_create.restype = P
    L.hbo_dense_create.argtypes = [U64, P, P, P, ctypes.c_int]
    L.hbo_dense_destroy.argtypes = [P]
    L.hbo_dense_step.restype = ctypes.c_int
    L.hbo_dense_step.argtypes = [P, ctypes.c_int, P]
    L.hbo_dense_step_local.argtypes = [P, ctypes.c_int]
    L.hbo_dense_pending_registers.restype = P
    L.hbo_dense_pending_registers.argtypes = [P]
    L.hbo_dense_step_finish.restype = ctypes.c_int
    L.hbo_dense_step_finish.argtypes = [P, ctypes.c_int, P]
    L.hbo_dense_run.restype = U64
    L.hbo_dense_run.argtypes = [P, ctypes.c_int]
    for f in ("hbo_dense_registers", "hbo_dense_kahan_sum", "hbo_dense_kahan_err", "hbo_dense_sizes"):
        getattr(L, f).restype = P
        getattr(L, f).argtypes = [P]
    L.hbo_dense_passes.restype = U64
    L.hbo_dense_passes.argtypes = [P]
    L.hbo_dense_finish.restype = U64
    L.hbo_dense_finish.argtypes = [P, P, P]
    L.hbo_dense_set_bsearch.argtypes = [P, ctypes.c_int]
    L.hbo_dense_state_hash.argtypes = [P, P]
    L.hbo_faithful_run.restype = U64
    L.hbo_faithful_run.argtypes = [P, U64, P, P, U64, P]
    L.hbo_faithful_run_pages.restype = U64
    L.hbo_faithful_run_pages.argtypes = [P, U64, P, U64, P, P, U64, P]
    L.hbo_bloom_num_bits.restype = U64
    L.hbo_bloom_num_bits.argtypes = [U64, ctypes.c_double]
    L.hbo_bloom_estimate_card.restype = U64
    L.hbo_bloom_estimate_card.argtypes = [U64, U64]
    _lib = L
    return L


def hll_add(reg, item):
    load().hbo_hll_add(reg.ctypes.data, item & 0xFFFFFFFFFFFFFFFF)


def hll_size(reg, variant=BSEARCH_RUST_1_82):
    reg = np.ascontiguousarray(reg, dtype=np.uint8)
    return load().hbo_hll_size_ex(reg.ctypes.data, variant, None, None)


def hll_sizes(regs, variant=BSEARCH_RUST_1_82):
    regs = np.ascontiguousarray(regs, dtype=np.uint8).reshape(-1, 64)
    L = load()
    out = np.zeros(len(regs), dtype=np.uint64)
    base = regs.ctypes.data
    for i in range(len(regs)):
        out[i] = L.hbo_hll_size_ex(base + 64 * i, variant, None, None)
    return out


def rank_results(vals):
    """harmonic_rank of results given in ascending NodeID order (store_harmonic, centrality/mod.rs:92-103)."""
    L = load()
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    out = np.zeros(len(vals), dtype=np.uint64)
    L.hbo_rank_results.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    L.hbo_rank_results.restype = ctypes.c_int
    assert L.hbo_rank_results(vals.ctypes.data if len(vals) else None, len(vals), out.ctypes.data if len(vals) else None) == 0
    return out


def kahan_sum(values):
    s, e = ctypes.c_double(0.0), ctypes.c_double(0.0)
    L = load()
    for v in values:
        L.hbo_kahan_add(ctypes.byref(s), ctypes.byref(e), float(v))
    return s.value, e.value


class Dense:
    """Dense-array HyperBall oracle on (id_low64, CSR by destination)."""

    def __init__(self, id_low64, row_ptr, src, threads=0):
        self.L = load()
        self.id_low64 = np.ascontiguousarray(id_low64, dtype=np.uint64)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.src = np.ascontiguousarray(src, dtype=np.uint32)
        self.n = len(self.id_low64)
        assert len(self.row_ptr) == self.n + 1
        self.h = self.L.hbo_dense_create(self.n, self.id_low64.ctypes.data, self.row_ptr.ctypes.data,
                                         self.src.ctypes.data if len(self.src) else None, threads)
        assert self.h

    def set_bsearch(self, variant):
        self.L.hbo_dense_set_bsearch(self.h, variant)

    def step(self, flags=FRONTIER):
        ps = PassStats()
        has = self.L.hbo_dense_step(self.h, flags, ctypes.byref(ps))
        return bool(has), dict(t=ps.pass_, active_edges=ps.active_edges, touched=ps.touched, changed=ps.changed)

    def step_local(self, flags=FRONTIER):
        self.L.hbo_dense_step_local(self.h, flags)

    def pending(self):
        """Writable (n, 64) view of the registers produced by step_local."""
        ptr = self.L.hbo_dense_pending_registers(self.h)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), (max(self.n, 1), 64))[:self.n]

    def step_finish(self, flags=FRONTIER):
        ps = PassStats()
        has = self.L.hbo_dense_step_finish(self.h, flags, ctypes.byref(ps))
        return bool(has), dict(t=ps.pass_, active_edges=ps.active_edges, touched=ps.touched, changed=ps.changed)

    def run(self, flags=FRONTIER):
        return self.L.hbo_dense_run(self.h, flags)

    def _view(self, ptr, ctype, shape):
        if self.n == 0:
            return np.zeros(shape, dtype=ctype)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape).copy()

    def registers(self):
        return self._view(self.L.hbo_dense_registers(self.h), ctypes.c_uint8, (self.n, 64))

    def kahan(self):
        return (self._view(self.L.hbo_dense_kahan_sum(self.h), ctypes.c_double, (self.n,)),
                self._view(self.L.hbo_dense_kahan_err(self.h), ctypes.c_double, (self.n,)))

    def sizes(self):
        return self._view(self.L.hbo_dense_sizes(self.h), ctypes.c_uint64, (self.n,))

    def passes(self):
        return self.L.hbo_dense_passes(self.h)

    def state_hash(self):
        """(registers checksum, Kahan checksum) of the state after the last executed pass."""
        out = np.zeros(2, dtype=np.uint64)
        self.L.hbo_dense_state_hash(self.h, out.ctypes.data)
        return int(out[0]), int(out[1])

    def finish(self):
        """(values[n], keep[n]) - normalize_centralities on index space."""
        out = np.zeros(max(self.n, 1), dtype=np.float64)
        keep = np.zeros(max(self.n, 1), dtype=np.uint8)
        k = self.L.hbo_dense_finish(self.h, out.ctypes.data, keep.ctypes.data)
        return out[:self.n], keep[:self.n].astype(bool), k

    def close(self):
        if self.h:
            self.L.hbo_dense_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def links_scorer(to_ids, self_id):
    """hbo_links_scorer: emit mask of one posting list (to_ids: U128 array in doc order; self_id: one U128 record)."""
    L = load()
    to_ids = np.ascontiguousarray(to_ids, dtype=U128)
    emit = np.zeros(max(len(to_ids), 1), dtype=np.uint8)
    L.hbo_links_scorer.restype = None
    L.hbo_links_scorer.argtypes = [ctypes.c_void_p, ctypes.c_uint64, _U128S, ctypes.c_void_p]
    me = _U128S(int(self_id["lo"]), int(self_id["hi"]))
    L.hbo_links_scorer(to_ids.ctypes.data if len(to_ids) else None, len(to_ids), me, emit.ctypes.data)
    return emit[:len(to_ids)].astype(bool)


def faithful_run(edges, pages=None, segments=None):
    """Structure-faithful single-thread path on raw SmallEdge records.  pages (EDGE records, may be empty): the
    sqrt(n) tail follows these page-level records like the reference (doc order, `segments` = segment
    lengths or None for one segment); None = host-level.
    Returns (ids[U128], vals[f64], stats dict)."""
    L = load()
    edges = np.ascontiguousarray(edges, dtype=EDGE)
    cap = 2 * len(edges) + 1
    ids = np.zeros(cap, dtype=U128)
    vals = np.zeros(cap, dtype=np.float64)
    st = FaithfulStats()
    if pages is None:
        k = L.hbo_faithful_run(edges.ctypes.data if len(edges) else None, len(edges), ids.ctypes.data, vals.ctypes.data,
                               cap, ctypes.byref(st))
    else:
        pages = np.ascontiguousarray(pages, dtype=EDGE)
        keep = np.zeros(1, dtype=EDGE)  # non-NULL even when there are no page records: "page-level, nothing found"
        seg = None if segments is None else np.ascontiguousarray(segments, dtype=np.uint64)
        L.hbo_faithful_run_segments.restype = ctypes.c_uint64
        L.hbo_faithful_run_segments.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                                ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        k = L.hbo_faithful_run_segments(edges.ctypes.data if len(edges) else None, len(edges),
                                        pages.ctypes.data if len(pages) else keep.ctypes.data, len(pages),
                                        seg.ctypes.data if seg is not None else None, 0 if seg is None else len(seg),
                                        ids.ctypes.data, vals.ctypes.data, cap, ctypes.byref(st))
    assert k != 0xFFFFFFFFFFFFFFFF
    return ids[:k].copy(), vals[:k].copy(), {f: getattr(st, f) for f, _ in st._fields_}
