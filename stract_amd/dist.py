"""Edge-partitioned multi-GPU driver: one process per GPU, RCCL over xGMI.

The reference has no multi-device form of this path (its distributed variant is a raft
DHT over TCP, crates/core/src/entrypoint/ampc/harmonic_centrality/); this is the new
design of SURVEY.md §8(e): every rank holds all n counters and 1/world of the edges; each
pass is  local pull-merge -> ncclAllReduce(max, u8) of the counters -> estimator + Kahan
on the rank's own node slice.  The collective itself is issued by the C library on its own
HIP stream (hb_api.hip); Python only partitions the input and distributes the
ncclUniqueId."""
import numpy as np

from . import _lib


def _mix64(x):
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xFF51AFD7ED558CCD)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xC4CEB9FE1A85EC53)
    x ^= x >> np.uint64(33)
    return x


def edge_owner(edges, world):
    """Rank that owns each SmallEdge record: a hash of the (from, to) pair, so that all
    records of one pair land on one rank in stream order and the reference's
    first-occurrence rule (store.rs:313) can be applied locally."""
    with np.errstate(over="ignore"):
        h = _mix64(edges["from"]["lo"] ^ _mix64(edges["from"]["hi"]))
        h = _mix64(h ^ _mix64(edges["to"]["lo"] + np.uint64(0x9E3779B97F4A7C15)) ^ _mix64(edges["to"]["hi"]))
    return (h % np.uint64(world)).astype(np.int64)


def partition_edges(edges, rank, world):
    """This rank's records, stream order preserved."""
    if world <= 1:
        return edges
    return edges[edge_owner(edges, world) == rank]


def partition_dense(row_ptr, src, rank, world):
    """This rank's share of a reduced graph: edge k of the CSR goes to rank k % world
    (every row's in-edges are spread over all ranks).  Returns (row_ptr, src) over ALL rows."""
    if world <= 1:
        return row_ptr, src
    # number of kept edges before position p: ceil((p - rank) / world) clipped at 0
    rp = row_ptr.astype(np.int64)
    before = np.maximum(0, (rp - rank + world - 1) // world)
    # (a strided view, copied once: no 8-byte index per edge - C4 has 2.1 G of them and every rank of `bench.py --gpus N` does this)
    return before.astype(np.uint64), np.ascontiguousarray(src[rank::world])


def partition_dense_by_dest(row_ptr, src, rank, world):
    """Destination partition (HB_FLAG_DEST_PARTITION): rank r gets ALL in-edges of the nodes whose
    index in ascending-NodeID order is r mod world, nothing else.  Returns (row_ptr, src) over ALL rows."""
    if world <= 1:
        return row_ptr, src
    rp = row_ptr.astype(np.int64)
    deg = np.diff(rp)
    mine = (np.arange(len(deg)) % world) == rank
    keep_deg = np.where(mine, deg, 0)
    out_rp = np.zeros(len(rp), dtype=np.uint64)
    np.cumsum(keep_deg, out=out_rp[1:])
    # edge mask: edges of owned rows
    edge_owner_row = np.repeat(mine, deg)
    return out_rp, np.ascontiguousarray(src[edge_owner_row])


def dest_owner_of_edges(edges, sorted_ids, world):
    """Owner rank of each raw SmallEdge record under the destination partition: the rank of its `to`
    id in the ascending node list, mod world (records may also simply be given to every rank: the
    library ignores records whose destination it does not own)."""
    to = edges["to"]
    key_ids = sorted_ids["hi"].astype(object) * (1 << 64) + sorted_ids["lo"].astype(object)
    key_to = to["hi"].astype(object) * (1 << 64) + to["lo"].astype(object)
    pos = np.searchsorted(key_ids, key_to)
    return (pos % world).astype(np.int64)


def torch_unique_id(rank, world):
    """Distribute rank 0's ncclUniqueId with torch.distributed (any backend)."""
    import torch
    import torch.distributed as td

    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(_lib.rccl_unique_id()), dtype=torch.uint8).clone()
    if td.get_backend() == "nccl":  # RCCL moves device memory only
        buf = buf.cuda()
    td.broadcast(buf, src=0)
    return bytes(buf.cpu().tolist())


def make_context(rank, world, device, rccl_id, **kw):
    return _lib.Context(device=device, rank=rank, world_size=world, rccl_id=rccl_id, **kw)


class HostStagedCollectives:
    """hb_set_collectives over torch.distributed tensors in HOST memory (any backend that moves CPU tensors: gloo): every
    exchange of the pass driver - all-reduce(max, u8) of the counters, the pipelined per-range form, the changed-only packing
    with its all-gather of bitmaps and per-rank broadcasts, the destination partition's all-gathers, the all-gather of the
    Kahan slices - runs as the LIBRARY's code, with N processes that may all sit on one device (RCCL refuses that).  Slow by
    construction (device -> host -> wire -> host -> device, blocking); its purpose is correctness of the multi-process
    protocol where only one GPU exists (tests/test_gpu.py), and as a template for integrators with their own fabric.

    Keep the object alive as long as the context uses it (it owns the ctypes callbacks)."""

    _ALL_REDUCE = __import__("ctypes").CFUNCTYPE(__import__("ctypes").c_int, __import__("ctypes").c_void_p, __import__("ctypes").c_void_p,
                                                 __import__("ctypes").c_uint64, __import__("ctypes").c_int, __import__("ctypes").c_int,
                                                 __import__("ctypes").c_void_p)
    _ALL_GATHER = __import__("ctypes").CFUNCTYPE(__import__("ctypes").c_int, __import__("ctypes").c_void_p, __import__("ctypes").c_void_p,
                                                 __import__("ctypes").c_void_p, __import__("ctypes").c_uint64, __import__("ctypes").c_void_p)
    _BROADCAST = __import__("ctypes").CFUNCTYPE(__import__("ctypes").c_int, __import__("ctypes").c_void_p, __import__("ctypes").c_void_p,
                                                __import__("ctypes").c_uint64, __import__("ctypes").c_int, __import__("ctypes").c_void_p)

    def __init__(self, ctx, group=None):
        import ctypes
        import torch
        import torch.distributed as td

        self.td, self.torch, self.group = td, torch, group
        self.world = td.get_world_size(group)
        self.rank = td.get_rank(group)
        self.lib = ctx.lib  # (the build that owns the context: product, or the experiments build when the context asked for one of its switches)
        self.calls = {"all_reduce": 0, "all_gather": 0, "broadcast": 0, "bytes": 0}
        self.error = None
        np_of = {_lib.HB_COLL_U8: np.uint8, _lib.HB_COLL_U32: np.uint32, _lib.HB_COLL_U64: np.uint64, _lib.HB_COLL_F64: np.float64}

        def to_host(dptr, nbytes, stream):
            host = np.empty(nbytes, dtype=np.uint8)
            if self.lib.hb_debug_staged_copy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dptr), nbytes, 0, ctypes.c_void_p(stream)) != _lib.HB_OK:
                raise RuntimeError("device -> host copy failed")
            return host

        def to_device(dptr, host, stream):
            host = np.ascontiguousarray(host)
            if self.lib.hb_debug_staged_copy(ctypes.c_void_p(dptr), host.ctypes.data_as(ctypes.c_void_p), host.nbytes, 1, ctypes.c_void_p(stream)) != _lib.HB_OK:
                raise RuntimeError("host -> device copy failed")

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # never unwind into the C library
                    self.error = repr(e)
                    return 1
            return wrapped

        @guard
        def all_reduce(user, buf, count, dtype, op, stream):
            dt = np.dtype(np_of[dtype])
            host = to_host(buf, count * dt.itemsize, stream).view(dt)
            if dtype == _lib.HB_COLL_U8:
                t = torch.from_numpy(host)                      # byte-wise max: the counters
            elif dtype == _lib.HB_COLL_F64:
                t = torch.from_numpy(host)
            else:                                               # u32 / u64 sums: as int64 (two's complement wrap = unsigned wrap)
                t = torch.from_numpy(host.astype(np.int64) if dtype == _lib.HB_COLL_U32 else host.view(np.int64))
            td.all_reduce(t, op=td.ReduceOp.MAX if op == _lib.HB_COLL_MAX else td.ReduceOp.SUM, group=self.group)
            out = t.numpy()
            if dtype == _lib.HB_COLL_U32:
                out = out.astype(np.uint32)
            to_device(buf, out.view(np.uint8) if out.dtype != np.uint8 else out, stream)
            self.calls["all_reduce"] += 1
            self.calls["bytes"] += int(count * dt.itemsize)

        @guard
        def all_gather(user, send, recv, nbytes, stream):
            mine = torch.from_numpy(to_host(send, nbytes, stream))
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            td.all_gather(parts, mine, group=self.group)
            to_device(recv, torch.cat(parts).numpy(), stream)
            self.calls["all_gather"] += 1
            self.calls["bytes"] += int(nbytes * self.world)

        @guard
        def broadcast(user, buf, nbytes, root, stream):
            t = torch.from_numpy(to_host(buf, nbytes, stream)) if self.rank == root else torch.empty(nbytes, dtype=torch.uint8)
            td.broadcast(t, src=root, group=self.group)
            if self.rank != root:
                to_device(buf, t.numpy(), stream)
            self.calls["broadcast"] += 1
            self.calls["bytes"] += int(nbytes)

        self._cb = (self._ALL_REDUCE(all_reduce), self._ALL_GATHER(all_gather), self._BROADCAST(broadcast))

        class Ops(ctypes.Structure):
            _fields_ = [("user", ctypes.c_void_p), ("all_reduce", self._ALL_REDUCE), ("all_gather", self._ALL_GATHER), ("broadcast", self._BROADCAST)]

        self._ops = Ops(None, *self._cb)
        rc = self.lib.hb_set_collectives(ctx.h, ctypes.byref(self._ops))
        if rc != _lib.HB_OK:
            raise _lib.HyperballError(rc, (self.lib.hb_last_error(ctx.h) or b"").decode())
