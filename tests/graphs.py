"""Shared test graphs (mirrors of the reference's fixtures + synthetic generators)."""
import numpy as np

from stract_amd import _lib
from stract_amd.harmonic import EdgeListGraph

# the reference fixture, crates/core/src/webgraph/centrality/harmonic.rs:323-341
#   A->B, B->C, A->C, C->A, D->C
A, B, C, D = 1, 2, 3, 4
FIXTURE = [(A, B), (B, C), (A, C), (C, A), (D, C)]
TAG = 1 << 13                # RelFlags::TAG, webpage/html/links.rs:130
SAME_ICANN_DOMAIN = 1 << 21  # links.rs:139
NOFOLLOW = 1 << 8


def fixture_graph(flags=0, extra=()):
    return EdgeListGraph.from_tuples([(f, t, flags) for f, t in FIXTURE] + list(extra))


def host_fixture():
    """harmonic.rs:358-458: twelve A.com->A.com page links collapse to one host self-loop;
    C.com->B.com, D.com->B.com."""
    a, b, c, d = 0xA0, 0xB0, 0xC0, 0xD0
    return EdgeListGraph.from_tuples([(a, a)] * 12 + [(c, b), (d, b)]), (a, b, c, d)


def lcg_graph(n=200, m=1200, seed=12345):
    """ids 1..n, m unique non-self edges from a 64-bit LCG."""
    x = seed
    edges = set()
    while len(edges) < m:
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        f = (x >> 33) % n + 1
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        t = (x >> 33) % n + 1
        if f != t:
            edges.add((f, t))
    return sorted(edges)


def dense_from_tuples(tuples):
    """(ids U128 ascending, row_ptr, src) of a clean tuple list (unique, no flags)."""
    nodes = sorted({x for e in tuples for x in e[:2]})
    index = {v: i for i, v in enumerate(nodes)}
    rows = [[] for _ in nodes]
    for f, t in sorted(set((e[0], e[1]) for e in tuples)):
        rows[index[t]].append(index[f])
    row_ptr = np.zeros(len(nodes) + 1, dtype=np.uint64)
    for i, r in enumerate(rows):
        row_ptr[i + 1] = row_ptr[i] + len(r)
    src = np.array([s for r in rows for s in sorted(r)], dtype=np.uint32)
    ids = np.zeros(len(nodes), dtype=_lib.U128)
    for i, v in enumerate(nodes):
        ids[i]["lo"] = v & 0xFFFFFFFFFFFFFFFF
        ids[i]["hi"] = v >> 64
    return ids, row_ptr, src


def random_registers(rng, count, kind="mixed"):
    """Random 64-register blocks covering every branch of HyperLogLog::size."""
    regs = np.zeros((count, 64), dtype=np.uint8)
    for i in range(count):
        k = kind if kind != "mixed" else ("sparse", "small", "mid", "large", "wide")[i % 5]
        if k == "sparse":      # many zero registers -> linear counting
            nz = rng.integers(1, 40)
            pos = rng.choice(64, nz, replace=False)
            regs[i, pos] = rng.integers(1, 6, nz)
        elif k == "small":     # e <= 320 -> bias table
            regs[i] = rng.integers(0, 4, 64)
        elif k == "mid":
            regs[i] = rng.integers(1, 8, 64)
        elif k == "large":
            regs[i] = rng.integers(8, 30, 64)
        else:                  # extreme register values incl. > 47 (sequential f64 fold)
            regs[i] = rng.integers(0, 66, 64)
    return regs


def random_graph(rng, kind=None):
    """Small random graphs of several shapes (as unique (from, to) int tuples, ids 1..n)."""
    kind = kind or ("uniform", "stars", "chains", "dense_core", "bipartite")[int(rng.integers(0, 5))]
    n = int(rng.integers(2, 600))
    edges = set()
    if kind == "uniform":
        for _ in range(int(rng.integers(0, 6 * n))):
            edges.add((int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))))
    elif kind == "stars":      # a few destinations with very many sources (deep chunk trees at small chunk sizes)
        for h in range(1, int(rng.integers(1, 5)) + 1):
            for s in rng.choice(np.arange(1, n + 1), size=int(rng.integers(1, n)), replace=False):
                edges.add((int(s), h))
        for _ in range(n):
            edges.add((int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))))
    elif kind == "chains":     # long diameter: many frontier / sparse passes
        for i in range(1, n):
            edges.add((i, i + 1))
        for _ in range(n // 4):
            edges.add((int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))))
    elif kind == "dense_core":
        core = max(2, n // 10)
        for a in range(1, core + 1):
            for b in range(1, core + 1):
                if rng.random() < 0.5:
                    edges.add((a, b))
        for v in range(core + 1, n + 1):
            edges.add((v, int(rng.integers(1, core + 1))))
            edges.add((int(rng.integers(1, core + 1)), v))
    else:                      # sources-only nodes -> destination-only nodes
        half = max(1, n // 2)
        for _ in range(4 * n):
            edges.add((int(rng.integers(1, half + 1)), int(rng.integers(half + 1, n + 2))))
    return kind, sorted(edges)


def tailed_graph(core=300, core_edges=1500, chain=120, seed=7, branch=3):
    """A graph that reaches the reference's sqrt(n) tail (harmonic.rs:244-252): an LCG core whose node 1 feeds a
    chain of `chain` hosts with a few side branches.  Ids are salted so that bloom slots are not trivially distinct.
    Returns host-level tuples (from, to, 0)."""
    salt = 0x9E3779B97F4A7C15
    def nid(k):
        return ((k * salt) & 0xFFFFFFFFFFFFFFFF) | (k << 64)
    e = [(nid(f), nid(t), 0) for f, t in lcg_graph(core, core_edges, seed)]
    first = core + 1
    e.append((nid(1), nid(first), 0))
    for k in range(chain - 1):
        e.append((nid(first + k), nid(first + k + 1), 0))
        if branch and k % branch == 0:
            e.append((nid(first + k), nid(first + chain + k), 0))  # a leaf hanging off the chain
    return e
