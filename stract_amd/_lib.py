"""ctypes binding of the C ABI declared in include/hyperball.h.

The shared library is built in-tree (stract_amd/lib/libhyperball.so) by
`__graft_entry__.build()` / `make -C stract_amd/csrc`.  There is no Python or CPU
fallback: if the library is missing, or no gfx950 device is present, calls raise.
"""
import ctypes
import os

# OpenMP worker threads of the host-side pieces (store writer, column reader, the synthetic-graph and oracle support
# libraries) must SLEEP when a parallel region ends, not spin: a box that shows 256 hardware threads under a 16-CPU
# container quota otherwise has hundreds of spinning workers competing with the one thread that drives the GPU - measured:
# ~100 ms lost in the first HIP wait after every parallel region (profiles/r04d_ingest_append_trace.txt).  libgomp reads the
# variable when it is loaded, i.e. before the libraries below are.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HB_LIB_PATH") or os.path.join(_HERE, "lib", "libhyperball.so")  # HB_LIB_PATH: debug / interpreter builds
# The EXPERIMENTS build (stract_amd/csrc/hb_experiments.h: A/B switches of rejected kernel forms, test hooks; `make exp`): the product
# library refuses those switches, so a Context that asks for one is served by this build.  Tests and measurement tools only.
LIB_EXP_PATH = os.environ.get("HB_LIB_PATH") or os.path.join(_HERE, "lib", "libhyperball_exp.so")

HB_OK = 0
HB_ERR_INVALID, HB_ERR_NO_DEVICE, HB_ERR_HIP, HB_ERR_NOMEM, HB_ERR_RCCL, HB_ERR_LIMIT, HB_ERR_IO = -1, -2, -3, -4, -5, -6, -7
HB_STORE_F64, HB_STORE_U64 = 0, 1
HB_COLL_U8, HB_COLL_U32, HB_COLL_U64, HB_COLL_F64 = 0, 1, 2, 3
HB_COLL_MAX, HB_COLL_SUM = 0, 1
HB_SKIPPED_REL_MASK = 0x6FED00

HB_FLAG_NO_FRONTIER = 0x01
HB_FLAG_NO_REORDER = 0x02
HB_FLAG_UNFUSED = 0x04
HB_FLAG_PASS_STATS = 0x08
HB_FLAG_NO_XCD_MAP = 0x20
HB_FLAG_NO_RCCL = 0x40
HB_FLAG_RCCL_SELF = 0x80
HB_FLAG_NO_SPARSE = 0x100
HB_FLAG_DEST_PARTITION = 0x200
HB_FLAG_HOST_INGEST = 0x400
HB_FLAG_HOST_PLAN = 0x800
HB_FLAG_CHANGED_ONLY = 0x1000
HB_FLAG_REFERENCE_TAIL = 0x2000
HB_FLAG_NO_INIT_PASS = 0x4000

# numpy views of the plain-data structs
U128 = np.dtype([("lo", "<u8"), ("hi", "<u8")])
EDGE = np.dtype([("from", U128), ("to", U128), ("rel_flags", "<u8")])
assert U128.itemsize == 16 and EDGE.itemsize == 40


class HbOptions(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("device", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("chunk", ctypes.c_uint32),
        ("max_passes", ctypes.c_uint32),
        ("rank", ctypes.c_int32),
        ("world_size", ctypes.c_int32),
        ("rccl_id", ctypes.c_uint8 * 128),
        ("tune", ctypes.c_uint32 * 8),
    ]


class HbStats(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64),
        ("m_input", ctypes.c_uint64),
        ("m_unique", ctypes.c_uint64),
        ("m_eff", ctypes.c_uint64),
        ("passes", ctypes.c_uint64),
        ("results", ctypes.c_uint64),
        ("ms_ingest", ctypes.c_double),
        ("ms_plan", ctypes.c_double),
        ("ms_h2d", ctypes.c_double),
        ("ms_loop", ctypes.c_double),
        ("ms_loop_gpu", ctypes.c_double),
        ("ms_collective", ctypes.c_double),
        ("ms_d2h", ctypes.c_double),
        ("work_rows", ctypes.c_uint64),
        ("virtual_rows", ctypes.c_uint64),
        ("device_bytes", ctypes.c_uint64),
        ("virtual_edges", ctypes.c_uint64),
        ("levels", ctypes.c_uint64),
        ("level1_edges", ctypes.c_uint64),
        ("level1_rows", ctypes.c_uint64),
        ("direct_edges", ctypes.c_uint64),
        ("rows_with_in_edges", ctypes.c_uint64),
        ("wire_bytes", ctypes.c_uint64),
        ("ingest_peak_bytes", ctypes.c_uint64),
        ("pool_peak_bytes", ctypes.c_uint64),
        ("result_stages", ctypes.c_uint64),
        ("result_list", ctypes.c_uint64),
        ("pipelined_passes", ctypes.c_uint64),
        ("tail_kernel_passes", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class HbPassStats(ctypes.Structure):
    _fields_ = [
        ("pass_", ctypes.c_uint64),
        ("changed", ctypes.c_uint64),
        ("active_edges", ctypes.c_uint64),
        ("touched", ctypes.c_uint64),
        ("mode", ctypes.c_uint32),
        ("ms_gpu", ctypes.c_float),
        ("ms_main", ctypes.c_float),
        ("ms_collective", ctypes.c_float),
        ("ms_level1", ctypes.c_float),
        ("reserved", ctypes.c_uint32),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["pass"] = d.pop("pass_")
        return d


# every symbol include/hyperball.h declares: (name, restype, argtypes)
_P = ctypes.c_void_p
_U64 = ctypes.c_uint64
_SIGNATURES = [
    ("hb_abi_version", ctypes.c_int, []),
    ("hb_release_cached_memory", ctypes.c_int, [ctypes.c_void_p]),
    ("hb_create", ctypes.c_int, [ctypes.POINTER(HbOptions), ctypes.POINTER(_P)]),
    ("hb_destroy", None, [_P]),
    ("hb_last_error", ctypes.c_char_p, [_P]),
    ("hb_load_edges", ctypes.c_int, [_P, _P, _U64, _P, _U64]),
    ("hb_append_edges", ctypes.c_int, [_P, _P, _U64]),
    ("hb_finalize", ctypes.c_int, [_P, _P, _U64]),
    ("hb_discard_appended", ctypes.c_int, [_P]),
    ("hb_load_tail_edges", ctypes.c_int, [_P, _P, _U64]),
    ("hb_append_tail_edges", ctypes.c_int, [_P, _P, _U64]),
    ("hb_tail_segment_end", ctypes.c_int, [_P]),
    ("hb_debug_set_ingest_limits", ctypes.c_int, [_P, _U64, _U64, _U64]),
    ("hb_set_collectives", ctypes.c_int, [_P, _P]),
    ("hb_debug_staged_copy", ctypes.c_int, [_P, _P, _U64, ctypes.c_int, _P]),
    ("hb_pinned_alloc", ctypes.c_int, [_U64, ctypes.POINTER(ctypes.c_void_p)]),
    ("hb_pinned_free", None, [ctypes.c_void_p]),
    ("hb_debug_h2d_rate", ctypes.c_int, [_P, _P, _U64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    ("hb_debug_tail_index", ctypes.c_int, [_U64, _P, _P, _U64, _P, _U64, _P, _P, _U64, ctypes.POINTER(ctypes.c_uint64)]),
    ("hb_load_dense", ctypes.c_int, [_P, _P, _U64, _P, _P, _U64]),
    ("hb_run", ctypes.c_int, [_P, ctypes.POINTER(HbStats)]),
    ("hb_begin", ctypes.c_int, [_P]),
    ("hb_step", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int)]),
    ("hb_finish", ctypes.c_int, [_P]),
    ("hb_get_stats", ctypes.c_int, [_P, ctypes.POINTER(HbStats)]),
    ("hb_get_pass_stats", ctypes.c_int, [_P, _U64, ctypes.POINTER(HbPassStats)]),
    ("hb_result_count", ctypes.c_int, [_P, ctypes.POINTER(_U64)]),
    ("hb_result_copy", ctypes.c_int, [_P, _P, _P, _U64]),
    ("hb_result_ranks", ctypes.c_int, [_P, _P, _U64]),
    ("hb_result_top", ctypes.c_int, [_P, _U64, _P, _P, ctypes.POINTER(ctypes.c_uint64)]),
    ("hb_rccl_unique_id", ctypes.c_int, [_P]),
    ("hb_device_count", ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ("hb_device_name", ctypes.c_int, [_P, ctypes.c_char_p, _U64]),
    ("hb_device_synchronize", ctypes.c_int, [_P]),
    ("hb_debug_copy_registers", ctypes.c_int, [_P, _P]),
    ("hb_debug_copy_kahan", ctypes.c_int, [_P, _P, _P]),
    ("hb_debug_copy_sizes", ctypes.c_int, [_P, _P]),
    ("hb_debug_hll_size", ctypes.c_int, [_P, _P, _U64, _P]),
    ("hb_debug_state_hash", ctypes.c_int, [_P, _P]),
    ("hb_debug_copy_graph", ctypes.c_int, [_P, _P, _P, _P]),
    ("hb_debug_copy_plan", ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    ("hb_debug_exchange", ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int]),
    ("hb_step_local", ctypes.c_int, [_P]),
    ("hb_step_finish", ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int)]),
    ("hb_host_ingest", ctypes.c_int, [_P, _U64, _P, _U64, ctypes.POINTER(_U64), ctypes.POINTER(_U64),
                                      ctypes.POINTER(_U64), _P, _P, _P]),
    ("hb_host_plan", ctypes.c_int, [_U64, _P, _P, ctypes.c_uint32, ctypes.c_uint32, _P, _P, _P, _P, _P, _P]),
]
# include/hb_webgraph.h
_SIGNATURES += [
    ("hbw_open", ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(_P)]),
    ("hbw_close", None, [_P]),
    ("hbw_last_error", ctypes.c_char_p, [_P]),
    ("hbw_num_segments", ctypes.c_int, [_P, ctypes.POINTER(_U64)]),
    ("hbw_segment_info", ctypes.c_int, [_P, _U64, ctypes.c_char_p, ctypes.POINTER(_U64)]),
    ("hbw_total_rows", ctypes.c_int, [_P, ctypes.POINTER(_U64)]),
    ("hbw_read_host_edges", ctypes.c_int, [_P, _U64, _U64, _P]),
    ("hbw_read_page_edges", ctypes.c_int, [_P, _U64, _U64, _P]),
    ("hb_load_webgraph", ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_uint32]),
    ("hbw_debug_sstable", ctypes.c_int, [_P, _U64, ctypes.c_int, _P, _U64, _P, _U64, ctypes.POINTER(_U64)]),
    ("hbw_debug_crc32", ctypes.c_uint32, [_P, _U64]),
]
# include/hb_ampc.h
_SIGNATURES += [
    ("hbu_create", ctypes.c_int, [ctypes.c_int32, _U64, ctypes.POINTER(_P)]),
    ("hbu_destroy", None, [_P]),
    ("hbu_last_error", ctypes.c_char_p, [_P]),
    ("hbu_len", ctypes.c_int, [_P, ctypes.POINTER(_U64)]),
    ("hbu_batch_set", ctypes.c_int, [_P, _P, _P, _U64]),
    ("hbu_batch_get", ctypes.c_int, [_P, _P, _U64, _P, _P]),
    ("hbu_batch_upsert", ctypes.c_int, [_P, _P, _P, _U64, _P]),
]
# include/hb_store.h
_SIGNATURES += [
    ("hb_store_write", ctypes.c_int, [ctypes.c_char_p, _P, _P, ctypes.c_int, _U64, ctypes.c_char_p, ctypes.c_size_t]),
    ("hb_store_harmonic", ctypes.c_int, [ctypes.c_char_p, _P, _P, _P, _U64, ctypes.c_char_p, ctypes.c_size_t]),
    ("hb_store_harmonic_results", ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_char_p, _U64]),
]
SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None
_lib_exp = None


def needs_experiments_build(tune):
    """hb_options.tune values that only the experiments build understands (hb_experiments.h): tune[1] above its low byte, tune[7]."""
    tune = tuple(tune)
    return (len(tune) > 1 and (int(tune[1]) & ~0xFF) != 0) or (len(tune) > 7 and int(tune[7]) != 0)


class HyperballError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hyperball error %d: %s" % (code, msg))
        self.code = code


def _open(path):
    if not os.path.exists(path):
        raise HyperballError(HB_ERR_INVALID,
                             "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C stract_amd/csrc` (there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    if hasattr(lib, "hb_simt_interpreter") and os.environ.get("HB_ALLOW_SIMT_INTERPRETER") != "1":
        # tests/simt builds the library's sources against a host interpreter of the device code: test infrastructure for
        # kernel LOGIC, never a way to compute.  Only tests/test_simt.py sets the variable (for its own child process).
        raise HyperballError(HB_ERR_NO_DEVICE, "%s is the SIMT-interpreter test build, not the gfx950 library: refused "
                             "(there is no CPU fallback)" % path)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load(experiments=False):
    """Load libhyperball.so (once).  Raises if it has not been built.  experiments=True: the experiments build (see LIB_EXP_PATH)."""
    global _lib, _lib_exp
    if experiments and LIB_EXP_PATH != LIB_PATH:
        if _lib_exp is None:
            _lib_exp = _open(LIB_EXP_PATH)
        return _lib_exp
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def device_count():
    n = ctypes.c_int(0)
    load().hb_device_count(ctypes.byref(n))
    return n.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class PinnedRecords:
    """A page-locked batch buffer of `count` SmallEdge records (hb_pinned_alloc = hipHostMalloc in the runtime the LIBRARY
    runs on): `.array` is a numpy view of it.  Pinned batches reach hb_append_edges at the full rate of the host link."""

    def __init__(self, count, dtype=None):
        self.dtype = np.dtype(EDGE if dtype is None else dtype)
        self.count = int(count)
        self._p = ctypes.c_void_p()
        nbytes = max(self.count, 1) * self.dtype.itemsize
        rc = load().hb_pinned_alloc(nbytes, ctypes.byref(self._p))
        if rc != HB_OK or not self._p.value:
            raise HyperballError(rc, "hb_pinned_alloc(%d bytes) failed" % nbytes)
        buf = (ctypes.c_uint8 * nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=self.count)

    def close(self):
        if self._p is not None and self._p.value:
            self.array = None
            load().hb_pinned_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def store_write(path, ids, values):
    """One speedy_kv database directory (include/hb_store.h): ids = U128 array, values = float64 (Db<NodeID, f64>) or
    uint64 (Db<NodeID, u64>) array of the same length.  Host only."""
    ids = np.ascontiguousarray(ids, dtype=U128)
    values = np.ascontiguousarray(values)
    if values.dtype == np.float64:
        kind = HB_STORE_F64
    elif values.dtype == np.uint64:
        kind = HB_STORE_U64
    else:
        raise TypeError("values must be float64 or uint64, not %s" % values.dtype)
    if len(ids) != len(values):
        raise ValueError("ids and values differ in length")
    err = ctypes.create_string_buffer(512)
    rc = load().hb_store_write(os.fsencode(path), _ptr(ids), _ptr(values), kind, len(ids), err, len(err))
    if rc != HB_OK:
        raise HyperballError(rc, err.value.decode(errors="replace"))


def store_harmonic(output, ids, centralities, ranks):
    """`<output>/harmonic` + `<output>/harmonic_rank`: what store_harmonic (centrality/mod.rs:72-114) leaves on disk."""
    ids = np.ascontiguousarray(ids, dtype=U128)
    centralities = np.ascontiguousarray(centralities, dtype=np.float64)
    ranks = np.ascontiguousarray(ranks, dtype=np.uint64)
    if not (len(ids) == len(centralities) == len(ranks)):
        raise ValueError("ids, centralities and ranks differ in length")
    err = ctypes.create_string_buffer(512)
    rc = load().hb_store_harmonic(os.fsencode(output), _ptr(ids), _ptr(centralities), _ptr(ranks), len(ids), err, len(err))
    if rc != HB_OK:
        raise HyperballError(rc, err.value.decode(errors="replace"))


class Context:
    """Thin owner of an hb_ctx*; methods map 1:1 onto the C entry points."""

    def __init__(self, device=-1, flags=0, chunk=0, max_passes=0, rank=0, world_size=1, rccl_id=None, tune=()):
        self.lib = load(experiments=needs_experiments_build(tune))
        opt = HbOptions()
        opt.struct_size = ctypes.sizeof(HbOptions)
        opt.device = device
        opt.flags = flags
        opt.chunk = chunk
        opt.max_passes = max_passes
        opt.rank = rank
        opt.world_size = world_size
        if rccl_id is not None:
            ctypes.memmove(opt.rccl_id, bytes(rccl_id), 128)
        for i, v in enumerate(tune):
            opt.tune[i] = int(v)
        h = ctypes.c_void_p()
        rc = self.lib.hb_create(ctypes.byref(opt), ctypes.byref(h))
        if rc != HB_OK:
            raise HyperballError(rc, (self.lib.hb_last_error(None) or b"").decode())
        self.h = h
        self._keep = []

    # -- plumbing
    def _check(self, rc):
        if rc != HB_OK:
            raise HyperballError(rc, (self.lib.hb_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.hb_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- input
    def load_edges(self, edges, node_ids=None):
        edges = np.ascontiguousarray(edges, dtype=EDGE)
        n = 0
        if node_ids is not None:
            node_ids = np.ascontiguousarray(node_ids, dtype=U128)
            n = len(node_ids)
        self._check(self.lib.hb_load_edges(self.h, _ptr(node_ids), n, _ptr(edges), len(edges)))

    def append_edges(self, edges):
        edges = np.ascontiguousarray(edges, dtype=EDGE)
        self._check(self.lib.hb_append_edges(self.h, _ptr(edges), len(edges)))

    def load_tail_edges(self, records):
        """HB_FLAG_REFERENCE_TAIL: the page-level records update_changed_counters follows (harmonic.rs:82-92)."""
        records = np.ascontiguousarray(records, dtype=EDGE)
        self._check(self.lib.hb_load_tail_edges(self.h, _ptr(records) if len(records) else None, len(records)))

    def append_tail_edges(self, records):
        records = np.ascontiguousarray(records, dtype=EDGE)
        self._check(self.lib.hb_append_tail_edges(self.h, _ptr(records) if len(records) else None, len(records)))

    def h2d_rate(self, host_array, reps=3):
        """GB/s at which this host buffer reaches the device through the library's stream (hb_debug_h2d_rate)."""
        g = ctypes.c_double(0.0)
        self._check(self.lib.hb_debug_h2d_rate(self.h, _ptr(host_array), host_array.nbytes, reps, ctypes.byref(g)))
        return g.value

    def set_ingest_limits(self, max_records=0, max_device_bytes=0, chunk_records=0):
        """Test hook (hb_debug_set_ingest_limits): reach the device ingest's refusal / spill / multi-chunk paths with small inputs."""
        self._check(self.lib.hb_debug_set_ingest_limits(self.h, max_records, max_device_bytes, chunk_records))

    def tail_segment_end(self):
        """The tail records appended since the last call were one whole segment of the store (doc order)."""
        self._check(self.lib.hb_tail_segment_end(self.h))

    def finalize(self, node_ids=None):
        n = 0
        if node_ids is not None:
            node_ids = np.ascontiguousarray(node_ids, dtype=U128)
            n = len(node_ids)
        self._check(self.lib.hb_finalize(self.h, _ptr(node_ids), n))

    def load_dense(self, sorted_ids, row_ptr, src):
        sorted_ids = np.ascontiguousarray(sorted_ids, dtype=U128)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        src = np.ascontiguousarray(src, dtype=np.uint32)
        assert len(row_ptr) == len(sorted_ids) + 1
        self._check(self.lib.hb_load_dense(self.h, _ptr(sorted_ids), len(sorted_ids), _ptr(row_ptr), _ptr(src), len(src)))

    # -- compute
    def run(self):
        st = HbStats()
        self._check(self.lib.hb_run(self.h, ctypes.byref(st)))
        return st.as_dict()

    def begin(self):
        self._check(self.lib.hb_begin(self.h))

    def step(self):
        has = ctypes.c_int(0)
        self._check(self.lib.hb_step(self.h, ctypes.byref(has)))
        return bool(has.value)

    def step_local(self):
        self._check(self.lib.hb_step_local(self.h))

    def step_finish(self):
        has = ctypes.c_int(0)
        self._check(self.lib.hb_step_finish(self.h, ctypes.byref(has)))
        return bool(has.value)

    @staticmethod
    def exchange(ctxs, phase=0):
        """Emulated collective between logical ranks on one device (hb_debug_exchange)."""
        arr = (ctypes.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        ctxs[0]._check(ctxs[0].lib.hb_debug_exchange(arr, len(ctxs), phase))

    def finish(self):
        self._check(self.lib.hb_finish(self.h))

    def stats(self):
        st = HbStats()
        self._check(self.lib.hb_get_stats(self.h, ctypes.byref(st)))
        return st.as_dict()

    def pass_stats(self):
        out = []
        t = 0
        while True:
            ps = HbPassStats()
            if self.lib.hb_get_pass_stats(self.h, t, ctypes.byref(ps)) != HB_OK:
                break
            out.append(ps.as_dict())
            t += 1
        return out

    def synchronize(self):
        self._check(self.lib.hb_device_synchronize(self.h))

    def device_name(self):
        buf = ctypes.create_string_buffer(64)
        self._check(self.lib.hb_device_name(self.h, buf, 64))
        return buf.value.decode()

    # -- results
    def results(self):
        k = ctypes.c_uint64(0)
        self._check(self.lib.hb_result_count(self.h, ctypes.byref(k)))
        ids = np.zeros(k.value, dtype=U128)
        vals = np.zeros(k.value, dtype=np.float64)
        self._check(self.lib.hb_result_copy(self.h, _ptr(ids), _ptr(vals), k.value))
        return ids, vals

    def ranks(self):
        """harmonic_rank of every result (store_harmonic order), aligned with results()."""
        k = ctypes.c_uint64(0)
        self._check(self.lib.hb_result_count(self.h, ctypes.byref(k)))
        out = np.zeros(k.value, dtype=np.uint64)
        self._check(self.lib.hb_result_ranks(self.h, _ptr(out), k.value))
        return out

    def store_harmonic(self, output):
        """store_harmonic (centrality/mod.rs:72-114) from the results this context holds; key order sorted on the device."""
        err = ctypes.create_string_buffer(512)
        rc = self.lib.hb_store_harmonic_results(self.h, os.fsencode(output), err, len(err))
        if rc != HB_OK:
            raise HyperballError(rc, err.value.decode(errors="replace"))

    def top(self, k):
        """top_nodes(TopNodes::Top(k)) (centrality/mod.rs:33-52): (ids, vals), largest centrality first."""
        cnt = ctypes.c_uint64(0)
        self._check(self.lib.hb_result_count(self.h, ctypes.byref(cnt)))
        k = min(int(k), cnt.value)
        ids = np.zeros(k, dtype=U128)
        vals = np.zeros(k, dtype=np.float64)
        w = ctypes.c_uint64(0)
        self._check(self.lib.hb_result_top(self.h, k, _ptr(ids), _ptr(vals), ctypes.byref(w)))
        return ids[:w.value], vals[:w.value]

    # -- debug exports
    def n(self):
        return self.stats()["n"]

    def registers(self):
        out = np.zeros((self.n(), 64), dtype=np.uint8)
        self._check(self.lib.hb_debug_copy_registers(self.h, _ptr(out)))
        return out

    def kahan(self):
        n = self.n()
        s = np.zeros(n, dtype=np.float64)
        e = np.zeros(n, dtype=np.float64)
        self._check(self.lib.hb_debug_copy_kahan(self.h, _ptr(s), _ptr(e)))
        return s, e

    def sizes(self):
        out = np.zeros(self.n(), dtype=np.uint64)
        self._check(self.lib.hb_debug_copy_sizes(self.h, _ptr(out)))
        return out

    def state_hash(self):
        """(registers checksum, Kahan checksum) of the current state (hb_debug_state_hash)."""
        out = np.zeros(2, dtype=np.uint64)
        self._check(self.lib.hb_debug_state_hash(self.h, _ptr(out)))
        return int(out[0]), int(out[1])

    def hll_size(self, regs):
        regs = np.ascontiguousarray(regs, dtype=np.uint8).reshape(-1, 64)
        out = np.zeros(len(regs), dtype=np.uint64)
        self._check(self.lib.hb_debug_hll_size(self.h, _ptr(regs), len(regs), _ptr(out)))
        return out

    def plan(self):
        """The device work layout as it lies in HBM (hb_debug_copy_plan), same dict as host_plan()."""
        sizes = np.zeros(4, dtype=np.uint64)
        self._check(self.lib.hb_debug_copy_plan(self.h, _ptr(sizes), None, None, None, None))
        n_pad, nv, slen, levels = (int(x) for x in sizes)
        order = np.zeros(n_pad, dtype=np.uint32)
        prp = np.zeros(n_pad + nv + 1, dtype=np.uint64)
        psrc = np.zeros(slen, dtype=np.uint32)
        lb = np.zeros(levels + 1, dtype=np.uint64)
        self._check(self.lib.hb_debug_copy_plan(self.h, _ptr(sizes), _ptr(order), _ptr(prp), _ptr(psrc), _ptr(lb)))
        return dict(order=order, row_ptr=prp, src=psrc, level_begin=lb, n_pad=n_pad, nv=nv)

    def graph(self):
        st = self.stats()
        ids = np.zeros(st["n"], dtype=U128)
        row_ptr = np.zeros(st["n"] + 1, dtype=np.uint64)
        src = np.zeros(st["m_eff"], dtype=np.uint32)
        self._check(self.lib.hb_debug_copy_graph(self.h, _ptr(ids), _ptr(row_ptr), _ptr(src)))
        return ids, row_ptr, src


def rccl_unique_id():
    buf = (ctypes.c_uint8 * 128)()
    rc = load().hb_rccl_unique_id(buf)
    if rc != HB_OK:
        raise HyperballError(rc, (load().hb_last_error(None) or b"").decode())
    return bytes(buf)


def host_ingest(edges, node_ids=None):
    """Host-only: the reference's node/edge-set semantics (no device needed).
    Returns (ids, row_ptr, src, m_unique)."""
    lib = load()
    edges = np.ascontiguousarray(edges, dtype=EDGE)
    nn = 0
    if node_ids is not None:
        node_ids = np.ascontiguousarray(node_ids, dtype=U128)
        nn = len(node_ids)
    n, mu, me = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib.hb_host_ingest(_ptr(node_ids), nn, _ptr(edges), len(edges), ctypes.byref(n), ctypes.byref(mu),
                            ctypes.byref(me), None, None, None)
    if rc != HB_OK:
        raise HyperballError(rc, (lib.hb_last_error(None) or b"").decode())
    ids = np.zeros(n.value, dtype=U128)
    row_ptr = np.zeros(n.value + 1, dtype=np.uint64)
    src = np.zeros(me.value, dtype=np.uint32)
    rc = lib.hb_host_ingest(_ptr(node_ids), nn, _ptr(edges), len(edges), ctypes.byref(n), ctypes.byref(mu),
                            ctypes.byref(me), _ptr(ids), _ptr(row_ptr), _ptr(src))
    if rc != HB_OK:
        raise HyperballError(rc, (lib.hb_last_error(None) or b"").decode())
    return ids, row_ptr, src, mu.value


def host_plan(row_ptr, src, flags=0, chunk=0, tune=()):
    """Host-only: the device work layout (order, plan_row_ptr, plan_src, level_begin, n_pad)."""
    lib = load()
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint32)
    n = len(row_ptr) - 1
    sizes = np.zeros(4, dtype=np.uint64)
    tn = np.zeros(8, dtype=np.uint32)
    tn[:len(tune)] = tune
    rc = lib.hb_host_plan(n, _ptr(row_ptr), _ptr(src), flags, chunk, _ptr(tn), _ptr(sizes), None, None, None, None)
    if rc != HB_OK:
        raise HyperballError(rc, (lib.hb_last_error(None) or b"").decode())
    n_pad, nv, slen, levels = (int(x) for x in sizes)
    order = np.zeros(n_pad, dtype=np.uint32)
    prp = np.zeros(n_pad + nv + 1, dtype=np.uint64)
    psrc = np.zeros(slen, dtype=np.uint32)
    lb = np.zeros(levels + 1, dtype=np.uint64)
    rc = lib.hb_host_plan(n, _ptr(row_ptr), _ptr(src), flags, chunk, _ptr(tn), _ptr(sizes), _ptr(order), _ptr(prp),
                          _ptr(psrc), _ptr(lb))
    if rc != HB_OK:
        raise HyperballError(rc, (lib.hb_last_error(None) or b"").decode())
    world = int(tn[7]) if tn[7] > 1 else 1
    if world == 1:
        order = order[:n]  # no padding rows inside: a permutation of the sids
    return dict(order=order, row_ptr=prp, src=psrc, level_begin=lb, n_pad=n_pad, nv=nv, slice=n_pad // world)
