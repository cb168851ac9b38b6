#!/usr/bin/env python3
"""Differential fuzzing of the pass driver against the oracle: random small graphs of several shapes x random layout / mode knobs,
compared after EVERY pass (registers, Kahan words, cached sizes, changed count, pass count) and at the end (final list).
Runs against whatever library HB_LIB_PATH names: the gfx950 library on a GPU box, or - on a machine without a GPU - the
interpreted test build of the device sources (tests/simt, with HB_ALLOW_SIMT_INTERPRETER=1; add the AddressSanitizer preload
for the `make asan` build).  A failure prints the seed and case that reproduce it and makes the exit code non-zero.

usage: tools/diff_fuzz.py [--mode passes|records|tail|ranks|mixed] [--seconds S] [--seed N] [--max-nodes N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import hbo  # noqa: E402
from stract_amd import _lib  # noqa: E402
from stract_amd.harmonic import EdgeListGraph  # noqa: E402
from tests import graphs  # noqa: E402

FLAG_POOL = ["NO_REORDER", "NO_XCD_MAP", "UNFUSED", "NO_SPARSE", "NO_FRONTIER", "PASS_STATS", "HOST_PLAN", "NO_INIT_PASS"]


def big_graph(rng, max_nodes):
    """Shapes the test-suite generator does not reach: hubs above 4096 in-edges (three-level chunk trees at small chunks), many
    isolated / source-only nodes, a heavy-tailed in-degree, long chains hanging off a dense core."""
    n = int(rng.integers(2, max_nodes))
    kind = ("heavy_tail", "mega_hub", "core_and_chains")[int(rng.integers(0, 3))]
    e = set()
    if kind == "heavy_tail":
        m = int(rng.integers(n, 8 * n))
        dst = np.minimum((rng.pareto(1.1, m) * 3).astype(np.int64) + 1, n)
        src = rng.integers(1, n + 1, m)
        e = set(zip(src.tolist(), dst.tolist()))
    elif kind == "mega_hub":
        k = int(rng.integers(1, 4))
        for h in range(1, k + 1):
            for s in rng.choice(np.arange(1, n + 1), size=int(rng.integers(n // 2, n)), replace=False).tolist():
                e.add((int(s), h))
        for _ in range(2 * n):
            e.add((int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))))
    else:
        core = max(2, n // 20)
        for a in range(1, core + 1):
            for b in rng.integers(1, core + 1, 8).tolist():
                e.add((a, int(b)))
        at = core + 1
        while at < n:
            length = int(rng.integers(1, 60))
            prev = int(rng.integers(1, core + 1))
            for v in range(at, min(at + length, n + 1)):
                e.add((prev, v))
                prev = v
            at += length
    return kind, sorted((a, b) for a, b in e if a != b)


def one_case(rng, max_nodes, case):
    if rng.random() < 0.5:
        kind, edges = graphs.random_graph(rng)
    else:
        kind, edges = big_graph(rng, max_nodes)
    if not edges:
        return None
    ids, row_ptr, src = graphs.dense_from_tuples(edges)
    chunk = int(rng.choice([4, 8, 16, 32, 64, 128, 256]))
    # (round 5 switches of tune[1]: 0x8000 = a result snapshot after every pass, 0x20000 = only the first one, 0x10000 = a 16-entry final list)
    tune = (int(rng.choice([0, 0, 1, 2, 7])), int(rng.choice([0, 1, 2, 4])) | int(rng.choice([0, 0, 0x100, 0x800, 0x2000])) | int(rng.choice([0, 0, 0x8000, 0x28000, 0x38000])) | int(rng.choice([0, 0, 0x200000, 0x400000])) | int(rng.choice([0, 0, 0x800000])) | int(rng.choice([0, 0, 0x2000000])),  # (+ the single-workgroup tail kernel: on / on after any pass; round 6: full init, transposition by scatter)
            int(rng.choice([0, 0, 30, 101])),
            int(rng.integers(4, 17)), int(rng.integers(1, 9)), int(rng.integers(0, chunk + 1)), int(rng.choice([0, 0, 1, 4, 1000000])))
    names = sorted(set(rng.choice(FLAG_POOL, size=int(rng.integers(0, 3))).tolist()))
    flags = 0
    for nm in names:
        flags |= getattr(_lib, "HB_FLAG_" + nm)
    what = dict(case=case, kind=kind, n=int(len(ids)), m=int(len(src)), chunk=chunk, tune=tune, flags=names)
    o = hbo.Dense(ids["lo"].copy(), row_ptr, src)
    with _lib.Context(flags=flags, chunk=chunk, tune=tune) as ctx:
        ctx.load_dense(ids, row_ptr, src)
        ctx.begin()
        assert np.array_equal(ctx.registers(), o.registers()), ("initial registers", what)
        has, t = True, 0
        while has:
            has = ctx.step()
            ohas, ost = o.step(hbo.FRONTIER)
            assert has == ohas, ("has_changes after pass %d" % t, what)
            assert np.array_equal(ctx.registers(), o.registers()), ("registers after pass %d" % t, what)
            s, e = ctx.kahan()
            os_, oe = o.kahan()
            assert np.array_equal(s.view(np.uint64), os_.view(np.uint64)) and np.array_equal(e.view(np.uint64), oe.view(np.uint64)), ("Kahan words after pass %d" % t, what)
            assert np.array_equal(ctx.sizes(), o.sizes()), ("sizes after pass %d" % t, what)
            assert ctx.pass_stats()[t]["changed"] == ost["changed"], ("changed count of pass %d" % t, what)
            t += 1
        ctx.finish()
        vals, keep, k = o.finish()
        gids, gvals = ctx.results()
        assert len(gvals) == k and np.array_equal(gids, ids[keep]) and np.array_equal(gvals.view(np.uint64), vals[keep].view(np.uint64)), ("final list", what)
    what["passes"] = t
    return what


SKIPPED_BITS = [8, 10, 11, 13, 14, 15, 16, 17, 18, 19, 21, 22]  # HB_SKIPPED_REL_MASK = 0x6FED00
HARMLESS_BITS = [0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 20]


def records_case(rng, max_nodes, case):
    """The record boundary: a random stream of SmallEdge records - 128-bit ids (equal low halves, different high halves among them),
    duplicates of a pair with other flags before and after it, skipped and harmless rel flags, self links - handed over in random
    batches (hb_append_edges ... hb_finalize, the device ingest: hash table of provisional ids, one stable sort) or at once, with and
    without an explicit node list; node / edge counts, pass count and the final list against the faithful (map-based) oracle."""
    n = int(rng.integers(2, max(3, max_nodes // 4)))
    m = int(rng.integers(1, 6 * n))
    pool_lo = rng.integers(1, 1 << 62, n, dtype=np.uint64)
    pool_hi = rng.integers(0, 3, n, dtype=np.uint64) if rng.random() < 0.5 else rng.integers(0, 1 << 63, n, dtype=np.uint64)
    if rng.random() < 0.3:
        pool_lo[: n // 2] = pool_lo[n // 2: n // 2 + n // 2]  # same low half, told apart only by the high half
    e = np.zeros(m, dtype=_lib.EDGE)
    a, b = rng.integers(0, n, m), rng.integers(0, n, m)
    if rng.random() < 0.5:  # a hub destination
        b[rng.random(m) < 0.3] = 0
    e["from"]["lo"], e["from"]["hi"], e["to"]["lo"], e["to"]["hi"] = pool_lo[a], pool_hi[a], pool_lo[b], pool_hi[b]
    flags = np.zeros(m, dtype=np.uint64)
    for k in np.nonzero(rng.random(m) < 0.25)[0]:
        bits = SKIPPED_BITS if rng.random() < 0.6 else HARMLESS_BITS
        flags[k] = np.uint64(1) << np.uint64(int(rng.choice(bits)))
    e["rel_flags"] = flags
    dup = np.nonzero(rng.random(m) < 0.2)[0]  # repeat some records elsewhere in the stream with other flags
    if len(dup):
        extra = e[dup].copy()
        extra["rel_flags"] = np.where(rng.random(len(dup)) < 0.5, np.uint64(0), np.uint64(1) << np.uint64(13))
        e = np.concatenate([e, extra])
        e = e[rng.permutation(len(e))]
    fids, fvals, fst = hbo.faithful_run(e)
    how = int(rng.integers(0, 3))
    what = dict(case=case, kind="records", records=int(len(e)), pool=n, how=("batches", "at once", "at once + node list")[how])
    flags_ctx = int(rng.choice([0, 0, _lib.HB_FLAG_HOST_INGEST, _lib.HB_FLAG_HOST_PLAN]))
    # hb_run: the tail pipeline (default) or one pass at a time (0x100000), results in snapshots (0x8000 / 0x38000) or at the end
    tune_ctx = (0, int(rng.choice([0, 0, 0x8000, 0x100000, 0x38000, 0x108000])) | int(rng.choice([0, 0, 0x200000, 0x400000])), int(rng.choice([0, 0, 101])), 0, 0, 0,
                int(rng.choice([0, 0, 1])))
    what["tune"] = tune_ctx
    with _lib.Context(flags=flags_ctx, chunk=int(rng.choice([8, 64])), tune=tune_ctx) as ctx:
        if how == 0:
            cuts = np.sort(rng.integers(0, len(e) + 1, int(rng.integers(0, 6))))
            for part in np.split(e, cuts):
                ctx.append_edges(part)
            ctx.finalize()
        elif how == 1:
            ctx.load_edges(e)
        else:
            nodes = np.unique(np.concatenate([e["from"], e["to"]]))
            ctx.load_edges(e, nodes)
        st = ctx.run()
        ids, vals = ctx.results()
    assert (st["n"], st["m_unique"], st["m_eff"], st["passes"]) == (fst["n"], fst["m_unique"], fst["m_eff"], fst["passes"]), ("counts", what, st, fst)
    assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64)), ("final list", what)
    what.update(m=int(fst["m_eff"]), passes=int(fst["passes"]))
    return what


def tail_case(rng, case):
    """HB_FLAG_REFERENCE_TAIL: the reference's changed-node machinery as written (bloom filter, exact-counting switch, sqrt(n) tail over
    page-level records replayed through the query's per-segment LinksScorer) on random tailed graphs (a core feeding a chain with side
    branches, so that 0 < |changed| <= sqrt(n) happens), with a random share of the host links present at page level, duplicates of
    documents under conflicting flags, foreign page ids, and random segment cuts; ids, values, pass count and the NUMBER OF TAIL PASSES
    against the faithful oracle given the same records."""
    core = int(rng.integers(20, 400))
    host = graphs.tailed_graph(core=core, core_edges=int(rng.integers(core, min(2000, core * (core - 1) // 3))), chain=int(rng.integers(5, 160)),
                               seed=int(rng.integers(1, 1 << 30)), branch=int(rng.integers(0, 5)))
    e = EdgeListGraph.from_tuples(host).host_edges()
    keep = rng.random(len(host)) < rng.choice([0.0, 0.3, 0.7, 1.0])
    pages = [host[i] for i in np.nonzero(keep)[0].tolist()]
    for i in np.nonzero(rng.random(len(host)) < 0.1)[0].tolist():  # duplicates of a document, some flagged, placed somewhere else
        f, t, _ = host[i]
        pages.insert(int(rng.integers(0, len(pages) + 1)), (f, t, int(rng.choice([0, graphs.NOFOLLOW, graphs.TAG]))))
    for k in range(int(rng.integers(0, 6))):  # documents whose endpoints are not hosts of the graph
        pages.append((0xDEAD0000 + k, host[int(rng.integers(0, len(host)))][1], 0))
    recs = EdgeListGraph.from_tuples(pages).host_edges() if pages else np.zeros(0, dtype=_lib.EDGE)
    cuts = sorted(set(int(x) for x in rng.integers(0, len(recs) + 1, int(rng.integers(0, 4)))) | {0, len(recs)})
    segs = [b - a for a, b in zip(cuts, cuts[1:])] or [0]
    fids, fvals, fst = hbo.faithful_run(e, recs, segs)
    what = dict(case=case, kind="tail", hosts=int(fst["n"]), host_edges=int(len(e)), page_records=int(len(recs)), segments=segs)
    with _lib.Context(flags=_lib.HB_FLAG_REFERENCE_TAIL | int(rng.choice([0, _lib.HB_FLAG_HOST_PLAN]))) as ctx:
        ctx.load_edges(e)
        at = 0
        for length in segs:
            for part in np.array_split(recs[at:at + length], int(rng.integers(1, 4))):
                ctx.append_tail_edges(part)
            ctx.tail_segment_end()
            at += length
        st = ctx.run()
        ids, vals = ctx.results()
        modes = [ps["mode"] for ps in ctx.pass_stats()]
    assert st["passes"] == fst["passes"] and modes.count(3) == fst["passes_exact"], ("pass / tail-pass count", what, modes, fst)
    assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64)), ("final list", what)
    what.update(m=int(len(e)), passes=int(fst["passes"]), tail_passes=int(fst["passes_exact"]))
    return what


def ranks_case(rng, max_nodes, case):
    """The multi-rank decompositions with R = 2..5 LOGICAL ranks on one device (the collective emulated by hb_debug_exchange): edge
    partition (all-reduce(max)), its changed-only form, destination partition (all-gather of the owned slices), its changed-only form;
    random graphs and layout knobs; after every pass all ranks must hold the oracle's registers, at the end the oracle's final list."""
    from stract_amd import dist
    kind, edges = graphs.random_graph(rng) if rng.random() < 0.5 else big_graph(rng, max(200, max_nodes // 3))
    if not edges:
        return None
    ids, row_ptr, src = graphs.dense_from_tuples(edges)
    world = int(rng.integers(2, 6))
    mode = str(rng.choice(["edge", "edge_changed", "dest", "dest_changed"]))
    flags = _lib.HB_FLAG_NO_RCCL | (_lib.HB_FLAG_DEST_PARTITION if mode.startswith("dest") else 0) | (_lib.HB_FLAG_CHANGED_ONLY if mode.endswith("_changed") else 0)
    chunk = int(rng.choice([8, 16, 64]))
    tune = (0, 0, int(rng.choice([0, 101])), int(rng.integers(4, 9)), int(rng.integers(1, 9)))
    what = dict(case=case, kind="ranks:" + kind, n=int(len(ids)), m=int(len(src)), world=world, mode=mode, chunk=chunk, tune=tune)
    o = hbo.Dense(ids["lo"].copy(), row_ptr, src)
    split = dist.partition_dense_by_dest if mode.startswith("dest") else dist.partition_dense
    ctxs = []
    try:
        for r in range(world):
            c = _lib.Context(rank=r, world_size=world, flags=flags, chunk=chunk, tune=tune)
            ctxs.append(c)
            rp, sr = split(row_ptr, src, r, world)
            c.load_dense(ids, rp, sr)
            c.begin()
        has, t = True, 0
        while has:
            for c in ctxs:
                c.step_local()
            _lib.Context.exchange(ctxs, 0)
            out = [c.step_finish() for c in ctxs]
            ohas, _ = o.step(hbo.FRONTIER)
            assert len(set(out)) == 1 and out[0] == ohas, ("has_changes after pass %d" % t, what)
            has = out[0]
            want = o.registers()
            for r, c in enumerate(ctxs):
                assert np.array_equal(c.registers(), want), ("registers of rank %d after pass %d" % (r, t), what)
            t += 1
        _lib.Context.exchange(ctxs, 1)
        vals, keep, k = o.finish()
        for r, c in enumerate(ctxs):
            c.finish()
            gids, gvals = c.results()
            assert len(gvals) == k and np.array_equal(gids, ids[keep]) and np.array_equal(gvals.view(np.uint64), vals[keep].view(np.uint64)), ("final list of rank %d" % r, what)
    finally:
        for c in ctxs:
            c.close()
    what["passes"] = t
    return what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["passes", "records", "tail", "ranks", "mixed"], default="passes")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-nodes", type=int, default=6000)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    done, edges, passes = 0, 0, 0
    case = 0
    while time.time() - t0 < a.seconds:
        try:
            if a.mode == "ranks" or (a.mode == "mixed" and case % 6 == 4):
                w = ranks_case(rng, a.max_nodes, case)
            elif a.mode == "tail" or (a.mode == "mixed" and case % 6 == 5):
                w = tail_case(rng, case)
            elif a.mode == "records" or (a.mode == "mixed" and case % 3 == 2):
                w = records_case(rng, a.max_nodes, case)
            else:
                w = one_case(rng, a.max_nodes, case)
        except AssertionError as e:
            print(json.dumps({"failed": str(e.args[0] if e.args else e), "seed": a.seed, "cases_before": done}))
            sys.exit(1)
        case += 1
        if w:
            done += 1
            edges += w["m"]
            passes += w["passes"]
    print(json.dumps({"library": _lib.LIB_PATH, "mode": a.mode, "seed": a.seed, "seconds": round(time.time() - t0, 1), "cases": done, "edges": edges, "passes_compared": passes, "failed": None}))


if __name__ == "__main__":
    main()
