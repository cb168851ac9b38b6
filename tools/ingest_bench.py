#!/usr/bin/env python3
"""The REAL boundary at BASELINE sizes: raw 40-byte SmallEdge records (what the Rust shim's `graph.host_edges()` loop hands
over, webgraph/store.rs:297-314) streamed slab by slab through hb_append_edges -> hb_finalize (GPU ingest + device planner),
then the run, then parity: the final (NodeID, f64) list against the CPU oracle's dense run over the clean graph.

The record stream is stract_amd/csrc/hb_synth.cpp hbs_stream_*: salt 2 = the clean edges in pseudo-random order mixed with
flagged-first pairs (lost for good), their later clean copies and flagged duplicates of clean edges - a stream the reference
semantics reduce to EXACTLY the clean graph (tests/test_host.py::test_streamed_export_reduces_to_clean_graph proves that
against the structure-faithful oracle at small sizes), so hb_stats must report n, m_eff of the clean graph and the result must
equal the oracle's.  No 40 B x m host array ever exists: one pinned slab is refilled.

usage: tools/ingest_bench.py <config> [--salt 0|2] [--slab RECORDS] [--verify] [--store DIR] [--out FILE]
  --verify      run the CPU oracle to convergence and compare the final list (else: compare with the hb_load_dense path)
  --store DIR   additionally write the stream as an on-disk edge store (tests/tantivy_fixture.py, segments of 8 Mi documents)
                and load it with hb_load_webgraph (native column reader)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stract_amd import _lib, synth  # noqa: E402


def sig(ids, vals):
    return (len(vals), int(vals.view(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF), int(ids["lo"].sum() & 0xFFFFFFFFFFFFFFFF))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--salt", type=int, default=2)
    ap.add_argument("--slab", type=int, default=1 << 24)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--store", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--torch-first", action="store_true",
                    help="import torch BEFORE the library is loaded: the library then runs on the HIP runtime bundled with torch "
                         "(same soname) instead of /opt/rocm's - the situation of bench.py --gpus N > 1")
    a = ap.parse_args()
    if a.torch_first:
        import torch  # noqa: F401

    t0 = time.perf_counter()
    g, scale, label = synth.make_config(a.config)
    t_gen = time.perf_counter() - t0
    total = g.stream_len(a.salt)
    slab = min(a.slab, max(total, 1))
    pinned = _lib.PinnedRecords(slab)  # hb_pinned_alloc: hipHostMalloc in the runtime the library runs on
    buf = pinned.array
    ctx = _lib.Context()
    h2d_gbs = ctx.h2d_rate(buf)  # what the link gives this process: pinned H2D of one slab on the library's stream
    hip_runtime = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
    out = {"config": a.config, "label": label, "n": int(g.n), "m_clean": int(g.m), "salt": a.salt, "records": total,
           "record_GB": round(total * 40 / 1e9, 2), "slab_records": slab, "s_generate_graph": round(t_gen, 1),
           "pinned_h2d_GBs": round(h2d_gbs, 1), "hip_runtime": hip_runtime}
    s_fill = s_append = 0.0
    at = 0
    t_all = time.perf_counter()
    while at < total:
        t0 = time.perf_counter()
        k = g.stream_fill(buf, at, a.salt)
        t1 = time.perf_counter()
        ctx.append_edges(buf[:k])
        t2 = time.perf_counter()
        s_fill += t1 - t0
        s_append += t2 - t1
        at += k
    t0 = time.perf_counter()
    ctx.finalize()
    s_finalize = time.perf_counter() - t0
    s_all = time.perf_counter() - t_all
    st = ctx.stats()
    lost = g.stream_lost_pairs(a.salt)
    out["boundary"] = {
        "s_fill_slabs_host": round(s_fill, 2), "s_append_edges": round(s_append, 2), "s_finalize": round(s_finalize, 2),
        "append_GBs": round(total * 40 / s_append / 1e9, 2), "append_vs_pinned_link": round(total * 40 / s_append / 1e9 / h2d_gbs, 3),
        "records_per_s_library": round(total / (s_append + s_finalize)), "records_per_s_incl_host_fill": round(total / s_all),
        "ms_ingest_reduce": round(st["ms_ingest"], 1), "ms_plan": round(st["ms_plan"], 1), "ms_h2d_state": round(st["ms_h2d"], 1),
        "ingest_peak_device_bytes": int(st["ingest_peak_bytes"]), "peak_bytes_per_record": round(st["ingest_peak_bytes"] / max(total, 1), 2),
        "allocator_held_peak_bytes": int(st["pool_peak_bytes"]),
        "device_bytes_resident": int(st["device_bytes"]),
        "stats_ok": bool(st["n"] == g.n and st["m_input"] == total and st["m_eff"] == g.m and st["m_unique"] == g.m + lost),
        "n": int(st["n"]), "m_input": int(st["m_input"]), "m_unique": int(st["m_unique"]), "m_eff": int(st["m_eff"])}
    run = ctx.run()
    ids, vals = ctx.results()
    out["run"] = {"passes": int(run["passes"]), "ms_loop": round(run["ms_loop"], 2), "gteps": round(g.m * run["passes"] / run["ms_loop"] / 1e6, 2),
                  "results": int(len(vals))}
    mine = sig(ids, vals)
    ctx.close()
    if a.verify:
        from oracle import hbo
        t0 = time.perf_counter()
        o = hbo.Dense(g.id_low64(), g.row_ptr, g.src, threads=min(os.cpu_count() or 1, 64))
        T = o.run()
        ovals, keep, k = o.finish()
        same = (T == run["passes"] and k == len(vals) and np.array_equal(ids, g.ids[keep]) and
                np.array_equal(vals.view(np.uint64), ovals[keep].view(np.uint64)))
        out["parity"] = {"bit_exact": bool(same), "scope": "final (NodeID, f64) list (%d results, %d passes) vs the CPU oracle's dense run over the "
                         "clean graph the stream reduces to" % (k, T), "s_oracle": round(time.perf_counter() - t0, 1)}
        o.close()
    else:
        with _lib.Context() as c2:
            c2.load_dense(g.ids, g.row_ptr, g.src)
            c2.run()
            i2, v2 = c2.results()
        out["parity"] = {"bit_exact": bool(len(v2) == len(vals) and np.array_equal(i2, ids) and np.array_equal(v2.view(np.uint64), vals.view(np.uint64))),
                         "scope": "final list vs the hb_load_dense path on the clean graph (same device code; the oracle compare is --verify)"}
    if a.store:
        from stract_amd import webgraph
        from tests import tantivy_fixture as tf
        seg = 1 << 23
        t0 = time.perf_counter()
        segs = []
        for b in range(0, total, seg):
            part = np.zeros(min(seg, total - b), dtype=_lib.EDGE)
            g.stream_fill(part, b, a.salt)
            segs.append(part)
        tf.write_edge_store(a.store, segs, extra_columns=False)
        s_write = time.perf_counter() - t0
        del segs
        with _lib.Context() as c3:
            t0 = time.perf_counter()
            webgraph.load_webgraph(c3, a.store, verify_crc=True)
            s_load = time.perf_counter() - t0
            st3 = c3.stats()
            c3.run()
            i3, v3 = c3.results()
        out["store"] = {"segments": (total + seg - 1) // seg, "s_write_fixture": round(s_write, 1), "s_hb_load_webgraph": round(s_load, 2),
                        "records_per_s": round(total / s_load), "stats_ok": bool(st3["n"] == g.n and st3["m_eff"] == g.m and st3["m_input"] == total),
                        "same_result": sig(i3, v3) == mine}
    line = json.dumps(out)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
