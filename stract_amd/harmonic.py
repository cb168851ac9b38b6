"""Host-side mirror of the reference's operator interface for this path.

Reference: crates/core/src/webgraph/centrality/harmonic.rs:289-311

    pub struct HarmonicCentrality(BTreeMap<NodeID, f64>);
    impl HarmonicCentrality {
        pub fn calculate(graph: &Webgraph) -> Self
        pub fn get(&self, node: &NodeID) -> Option<f64>
        pub fn iter(&self) -> impl Iterator<Item = (&NodeID, f64)>   // ascending NodeID
        pub fn len(&self) -> usize
        pub fn is_empty(&self) -> bool
    }

Same names, argument meaning and error behaviour; a key is absent exactly when its
centrality is <= 0.  `graph` is anything exposing what the Rust shim takes from `&Webgraph`
(webgraph/mod.rs:157,192): `host_nodes()` -> array of NodeID and `host_edges()` -> array of
SmallEdge records.  All arithmetic happens on the GPU behind the C ABI (stract_amd._lib);
there is no CPU path.
"""
import numpy as np

from . import _lib


def node_id(value):
    """NodeID from a Python int (u128)."""
    a = np.zeros((), dtype=_lib.U128)
    a["lo"] = value & 0xFFFFFFFFFFFFFFFF
    a["hi"] = value >> 64
    return a


def ids_from_ints(values):
    a = np.zeros(len(values), dtype=_lib.U128)
    for i, v in enumerate(values):
        a[i]["lo"] = v & 0xFFFFFFFFFFFFFFFF
        a[i]["hi"] = v >> 64
    return a


def ids_to_ints(ids):
    return [(int(h) << 64) | int(l) for l, h in zip(ids["lo"].tolist(), ids["hi"].tolist())]


class EdgeListGraph:
    """Minimal stand-in for `&Webgraph`: a list of SmallEdge records in stream order."""

    def __init__(self, edges):
        self._edges = np.ascontiguousarray(edges, dtype=_lib.EDGE)

    @classmethod
    def from_tuples(cls, tuples):
        """tuples: (from_int, to_int[, rel_flags])"""
        e = np.zeros(len(tuples), dtype=_lib.EDGE)
        for i, t in enumerate(tuples):
            f, to = t[0], t[1]
            e[i]["from"]["lo"] = f & 0xFFFFFFFFFFFFFFFF
            e[i]["from"]["hi"] = f >> 64
            e[i]["to"]["lo"] = to & 0xFFFFFFFFFFFFFFFF
            e[i]["to"]["hi"] = to >> 64
            e[i]["rel_flags"] = t[2] if len(t) > 2 else 0
        return cls(e)

    def host_nodes(self):
        """webgraph/mod.rs:157 -> store.rs:338-357: unique endpoints of all records."""
        both = np.concatenate([self._edges["from"], self._edges["to"]])
        return np.unique(both)

    def host_edges(self):
        """webgraph/mod.rs:192 -> store.rs:297-314 (de-duplication happens in the library)."""
        return self._edges


class HarmonicCentrality:
    def __init__(self, ids, vals, stats=None, pass_stats=None, ranks=None):
        self._ids = ids
        self._vals = vals
        self._ranks = ranks  # harmonic_rank of every result (hb_result_ranks), filled by calculate()
        self._map = None
        self.stats = stats or {}
        self.pass_stats = pass_stats or []

    @classmethod
    def calculate(cls, graph, **ctx_kwargs):
        """harmonic.rs:292.  Raises HyperballError if the GPU library is unavailable."""
        with _lib.Context(**ctx_kwargs) as ctx:
            ctx.load_edges(graph.host_edges(), graph.host_nodes())
            st = ctx.run()
            ids, vals = ctx.results()
            return cls(ids, vals, st, ctx.pass_stats(), ctx.ranks())

    @classmethod
    def calculate_dense(cls, sorted_ids, row_ptr, src, **ctx_kwargs):
        """Same computation on a pre-reduced graph (hb_load_dense)."""
        with _lib.Context(**ctx_kwargs) as ctx:
            ctx.load_dense(sorted_ids, row_ptr, src)
            st = ctx.run()
            ids, vals = ctx.results()
            return cls(ids, vals, st, ctx.pass_stats(), ctx.ranks())

    def _ensure_map(self):
        if self._map is None:
            self._map = dict(zip(ids_to_ints(self._ids), self._vals.tolist()))
        return self._map

    def get(self, node):
        """harmonic.rs:296: Some(centrality) or None."""
        if not isinstance(node, int):
            node = (int(node["hi"]) << 64) | int(node["lo"])
        return self._ensure_map().get(node)

    def iter(self):
        """harmonic.rs:300: (NodeID, centrality) in ascending NodeID order."""
        return zip(ids_to_ints(self._ids), self._vals.tolist())

    def arrays(self):
        return self._ids, self._vals

    def ranks(self):
        """Position of every result in the order store_harmonic ranks by (centrality descending by f64::total_cmp, then
        NodeID ascending; centrality/mod.rs:92-103), aligned with arrays()."""
        return self._ranks

    def len(self):
        return len(self._vals)

    __len__ = len

    def is_empty(self):
        return self.len() == 0


def store_harmonic(centralities, output):
    """centrality/mod.rs:72-114 `store_harmonic(centralities, output)`: the `harmonic` (NodeID -> f64) and `harmonic_rank`
    (NodeID -> u64) speedy_kv databases under `output`, written natively (include/hb_store.h; format unpinned) from the arrays
    the library returned - ranks come from hb_result_ranks (GPU radix sort), nothing is recomputed on the host.
    `centralities`: a HarmonicCentrality from calculate() / calculate_dense()."""
    ids, vals = centralities.arrays()
    ranks = centralities.ranks()
    if ranks is None:
        raise ValueError("this HarmonicCentrality carries no ranks (it was not produced by calculate())")
    _lib.store_harmonic(output, ids, vals, ranks)
