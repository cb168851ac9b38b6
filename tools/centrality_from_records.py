#!/usr/bin/env python3
"""Offline bridge for integrators without the FFI shim: harmonic centrality of a host graph dumped as raw
SmallEdge records.

Input : a binary file of 40-byte records {from: u128 LE, to: u128 LE, rel_flags: u64 LE} in the order
        `Webgraph::host_edges()` yields them (crates/core/src/webgraph/mod.rs:192) - e.g. written from Rust with
        `for e in graph.host_edges() { w.write_all(&e.from.as_u128().to_le_bytes())?; ... }`.
Output: CSV `node_id_hex,centrality,rank` in ascending NodeID order = HarmonicCentrality::iter()
        (harmonic.rs:300) plus the harmonic_rank store_harmonic would write (centrality/mod.rs:92-103).

usage: centrality_from_records.py <records.bin> <out.csv> [--chunk-records N]
Needs an MI355X (no CPU fallback)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stract_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("records")
    ap.add_argument("out")
    ap.add_argument("--chunk-records", type=int, default=1 << 24, help="records per hb_append_edges call")
    a = ap.parse_args()
    size = os.path.getsize(a.records)
    if size % _lib.EDGE.itemsize:
        sys.exit("%s: size %d is not a multiple of the 40-byte record" % (a.records, size))
    edges = np.memmap(a.records, dtype=_lib.EDGE, mode="r")
    with _lib.Context() as ctx:
        for o in range(0, len(edges), a.chunk_records):
            ctx.append_edges(np.ascontiguousarray(edges[o:o + a.chunk_records]))
        ctx.finalize()
        st = ctx.run()
        ids, vals = ctx.results()
        ranks = ctx.ranks()
    with open(a.out, "w") as f:
        f.write("node_id_hex,centrality,rank\n")
        for lo, hi, v, r in zip(ids["lo"].tolist(), ids["hi"].tolist(), vals.tolist(), ranks.tolist()):
            f.write("%032x,%r,%d\n" % ((hi << 64) | lo, v, r))
    print("n=%d m_unique=%d m_eff=%d passes=%d results=%d loop=%.1f ms" %
          (st["n"], st["m_unique"], st["m_eff"], st["passes"], len(vals), st["ms_loop"]), file=sys.stderr)


if __name__ == "__main__":
    main()
