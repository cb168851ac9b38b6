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
    m = len(src)
    mine = np.arange(rank, m, world, dtype=np.int64)
    # number of kept edges before position p: ceil((p - rank) / world) clipped at 0
    rp = row_ptr.astype(np.int64)
    before = np.maximum(0, (rp - rank + world - 1) // world)
    return before.astype(np.uint64), np.ascontiguousarray(src[mine])


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
