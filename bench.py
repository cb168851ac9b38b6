#!/usr/bin/env python3
"""bench.py - HyperBall (webgraph harmonic centrality) hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by
torch.distributed.run, one rank per GPU).  Prints ONE JSON line on rank 0.

  step      = one complete HyperBall run over the resident graph: initialize + every pass of
              the loop harmonic.rs:237-280 until a pass changes nothing (T passes, the last
              one being the reference's no-change pass) + normalize + result download.
  metric    = traversed edges per second: m_eff * T * K / t   (SURVEY.md §8(d)), the graph
              (CSR by destination) already resident in HBM when the timed region starts.
  workload  = BASELINE.json configs[3] (100M-host / 2B-edge R-MAT scale 28: the graph the north-star target is quoted on; it
              fits one GPU) by default [round 5; rounds 1-4: configs[2]]; --config C3 = configs[2] (10M / 200M), C2 = configs[1]
              (1M/20M: fits the 256 MiB Infinity Cache, so it says little about HBM), C5 = configs[4], LT = the long-tail graph.
  input     = N = 1: the graph enters through the drop-in boundary - raw 40-byte SmallEdge records (with
              flagged-first pairs and duplicates, a stream the reference semantics reduce to exactly the
              clean graph) streamed from a page-locked batch buffer (hb_pinned_alloc) through hb_append_edges +
              hb_finalize (detail.input: records/s, link rate, peak device bytes); --input dense = the bench-only
              hb_load_dense export.  PyTorch is imported for N > 1 only (torch.distributed); at N = 1 the process
              holds no torch and the library runs on /opt/rocm's HIP runtime.
  N > 1     = the same graph partitioned over the ranks (strong scaling), one RCCL collective of
              the counters per pass.  `value` = the north-star decomposition: edge
              partition + ncclAllReduce(max, u8); with --partition both (default) its changed-only form, the
              destination partition (+ ncclAllGather) and its changed-only variant run as extra legs under
              detail.partitions, each with GTEPS, ms_collective, wire bytes and a same-result field.
              --collectives host-staged: the same N > 1 path as a FUNCTIONAL run where N GPUs do not exist (gloo, host-staged
              exchanges through hb_set_collectives, ranks share the devices there are); the line says so, its numbers are no measurement.
  c3 leg    = with the default config at N = 1 the line also carries detail.c3: BASELINE configs[2] (10M hosts / 200M edges, the
              headline of rounds 1-4) - GTEPS, roofline fractions, parity, its own end-to-end chain (~1.5 min, a child process).
              (--c4-leg on: the same for configs[3] under detail.c4 when another config is the main one.)
  end to end= N = 1, record input (default): detail.end_to_end (and detail.c4.end_to_end) = the reference command's whole chain
              on the same graph (entrypoint/centrality.rs:41-71): an on-disk edge store (written by the harness, untimed) ->
              hb_load_webgraph (CRC-32 checked, native column reader, GPU ingest) -> hb_run -> hb_store_harmonic_results (the result
              list, the ranks and the key order of both speedy_kv databases from the device, then the files), seconds per stage, records/s of the load, entries/s of the
              store emission, compute share of the total, and three checks (graph = clean graph, result = the record leg's,
              a sample of keys read back from the written databases); a failed check makes the exit code non-zero.
  roofline  = SURVEY.md §8(d): HBM-bound.  Headline `achieved`/`frac` = the WHOLE generic dense pass
              (dense passes t >= 1: B_t with their own A_t) over its measured GPU time (HIP events on
              the library's stream) vs 8 TB/s.  Pass 0 (68*m_eff + 192.25*n algorithmic bytes) is
              reported apart in `pass0`: the library streams the sources' single initial registers
              (2 B per edge) there instead of gathering counters, so it must not lift the headline.
              `dominant_kernel` = the level-1 hub-chunk launch alone: 68 B x the REAL edges it
              gathers (no partial-row traffic booked), over its own event-timed duration.
              `whole_loop` = sum_t B_t / t_loop with B_t = 68*A_t + 4*(m-A_t) + 184*V_t + 8n + n/4.
  parity    = every line carries `parity_bit_exact`.  N = 1: the oracle always runs to CONVERGENCE (C4: ~10 passes, ~1 min on the
              box's host cores) and the final (NodeID, f64) list is compared - the CPU-baseline SAMPLE is the first passes that fit
              --cpu-seconds, the passes after that are parity work, not timed.  N > 1 (and --parity budget): final list when the CPU
              run converges inside its budget, else an order-independent checksum of all registers + Kahan state after the last
              pass the CPU finished.
  process   = the measurement runs in THIS process; any failure of the product path - hb_append_edges / hb_finalize not
              reducing the record stream to the clean graph, a HIP error, a GPU fault that aborts the process - ends the
              bench with a non-zero exit code and no JSON line (no fallback to hb_load_dense, no restart: round 3 had both
              and the review rightly called them nets under a broken boundary).  Only the C4 leg is a child process (the
              2 B-edge generator needs tens of GB of host memory); its failure is reported in detail.c4.error AND makes
              the exit code non-zero.  N > 1: a watchdog thread around the extra partition legs (--legs-timeout).
  cpu_baseline = the CPU oracle's dense OpenMP port of the reference arithmetic, timed on
              this box's host cores on the same graph for a bounded number of passes; the GPU time
              of the SAME passes is reported next to it (like for like); `cpu_faithful` = the
              single-thread structure-faithful form on C1.
"""
import os

# before anything that may load an OpenMP runtime: idle OpenMP workers sleep instead of spinning (see stract_amd/_lib.py)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import argparse  # noqa: E402
import json  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("HB_BENCH_CONFIG", "C4"), help="C1|C2|C3|C4|C5|LT or scale:m (default C4 = BASELINE configs[3])")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline time bound (0 = skip)")
    ap.add_argument("--input", default=os.environ.get("HB_BENCH_INPUT", "records"), choices=["records", "dense"],
                    help="N = 1: how the graph enters the library.  records (default) = the drop-in boundary: raw 40-byte SmallEdge "
                         "records streamed through hb_append_edges + hb_finalize (GPU ingest, device planner); dense = the bench-only "
                         "hb_load_dense export (pre-reduced CSR).  N > 1 always uses dense (every rank slices the same CSR)")
    ap.add_argument("--partition", default="both", choices=["both", "edge", "dest"],
                    help="N > 1: edge = the north-star edge partition + ncclAllReduce(max,u8) per pass (this is `value`); dest = "
                         "destination partition + ncclAllGather per pass; both (default) = edge first, then dest and dest+changed-only "
                         "as extra legs under detail.partitions")
    ap.add_argument("--changed-only", action="store_true",
                    help="N > 1: the main leg exchanges only the counters that changed (HB_FLAG_CHANGED_ONLY; edge partition: all-reduce over "
                         "the union of the locally changed rows)")
    ap.add_argument("--c4-leg", default="off", choices=["auto", "on", "off"],
                    help="append a BASELINE configs[3] (100M-host / 2B-edge) leg under detail.c4 (on: when another config is the main one; "
                         "auto = off: C4 IS the default config since round 5)")
    ap.add_argument("--c3-leg", default="auto", choices=["auto", "on", "off"],
                    help="append a BASELINE configs[2] (10M-host / 200M-edge) leg under detail.c3 (auto: with the default config C4 at N = 1)")
    ap.add_argument("--parity", default="auto", choices=["auto", "full", "budget"],
                    help="full: the oracle runs to convergence whatever --cpu-seconds says (the CPU-baseline sample stays the passes inside "
                         "the budget) and the FINAL LIST is compared; budget: stop the oracle with the budget (checksum of the state at that "
                         "pass).  auto = full at N = 1, budget at N > 1")
    ap.add_argument("--end-to-end", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: also run the reference command's whole chain on the same graph (entrypoint/centrality.rs:41-71): on-disk edge "
                         "store -> hb_load_webgraph -> hb_run -> results + ranks -> hb_store_harmonic, seconds per stage under "
                         "detail.end_to_end (auto: with record input at N = 1 for C3 / C4 / LT)")
    ap.add_argument("--e2e-dir", default="", help="where the end-to-end leg puts its edge store and output databases "
                                                  "(default: $TMPDIR or /tmp if the store fits on that disk, else /dev/shm)")
    ap.add_argument("--no-supervisor", action="store_true", help="accepted and ignored (round 3 measured in a restartable child process)")
    ap.add_argument("--legs-timeout", type=int, default=300,
                    help="N > 1, --partition both: seconds the extra partition legs may take before the line is printed without them")
    ap.add_argument("--collectives", default="rccl", choices=["rccl", "host-staged"],
                    help="N > 1 only.  rccl (default): one rank per GPU, the library's own RCCL exchanges = the measurement.  host-staged: "
                         "a FUNCTIONAL run of the same N > 1 code path where N GPUs do not exist - torch.distributed over gloo, the per-pass "
                         "exchanges through hb_set_collectives on host-staged buffers (stract_amd.dist.HostStagedCollectives), ranks share "
                         "the devices there are; the line says so and its numbers are not a measurement")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--tune", default="", help="comma separated hb_options.tune values")
    ap.add_argument("--verify", action="store_true", help="run the CPU oracle to convergence and compare the final result")
    ap.add_argument("--pass-log", default="", help="write per-pass stats JSON here")
    return ap.parse_args()


def pass_bytes(ps, n, m_eff, rows_with_in):
    """Algorithmic HBM bytes of one pass, SURVEY.md §8(d):
    B_t = 68*A_t + 4*(m_eff - A_t) + 184*V_t + 8*n + n/4; pass 0 = 68*m_eff + 192.25*n."""
    a = min(int(ps["active_edges"]), m_eff)
    if ps["pass"] == 0:
        a, v = m_eff, n
    elif ps["mode"] == 0:
        v = rows_with_in  # dense pass: every row with an in-edge has (almost surely) an active one
    else:
        v = int(ps["touched"])
    return 68.0 * a + 4.0 * (m_eff - a) + 184.0 * v + 8.0 * n + n / 4.0


def moved_bytes(ps, n, m_eff, rows_with_in, work_rows, init_streamed):
    """Bytes the pass actually has to move in THIS implementation (a model, next to the §8(d) bytes it is booked with):
    pass 0 streams 2 B per edge instead of gathering; a sweep pass (mode 2) never reads the index lists of untouched rows
    (no 4*(m - A_t) term, no per-node row pointers) but reads/clears the touch bitmap and three per-node bitmaps."""
    a = min(int(ps["active_edges"]), m_eff)
    if ps["pass"] == 0 and init_streamed:
        return 2.0 * m_eff + 4.0 * m_eff + 192.25 * n  # initial registers + indices + the node state
    if ps["mode"] == 2:
        return 68.0 * a + 184.0 * int(ps["touched"]) + work_rows / 4.0 + 3.0 * n / 8.0
    return pass_bytes(ps, n, m_eff, rows_with_in)


def measure(ctx, steps, warmup, barrier, td, torch):
    """W untimed + K timed complete runs (hb_begin + all passes + hb_finish); wall time bracketed by barrier + synchronize, max over ranks."""
    r = {"loop_ms": 0.0, "gpu_ms": 0.0, "coll_ms": 0.0, "d2h_ms": 0.0, "passes": 0, "pass_stats": []}
    # the FIRST run of a freshly loaded context, timed on its own (wall clock around the one blocking call): it is the only run
    # `stract centrality harmonic` ever makes (entrypoint/centrality.rs:49), so it is reported next to the steady-state ms_per_step
    barrier()
    t0 = time.perf_counter()
    if warmup > 0:
        ctx.run()
        r["first_run_ms"] = (time.perf_counter() - t0) * 1e3
        r["first_run_finish_ms"] = ctx.stats()["ms_d2h"]
    for _ in range(max(warmup - 1, 0)):
        ctx.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run()
        st = ctx.stats()
        r["passes"] = int(st["passes"])
        r["loop_ms"] += st["ms_loop"]
        r["gpu_ms"] += st["ms_loop_gpu"]
        r["coll_ms"] += st["ms_collective"]
        r["d2h_ms"] += st["ms_d2h"]
        r["pass_stats"].append(ctx.pass_stats())
    barrier()
    dt = time.perf_counter() - t0
    if td is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if td.get_backend() == "nccl" else "cpu")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    r["dt"] = dt
    r["stats"] = ctx.stats()
    return r


def load_records(ctx, g, salt=2, slab=1 << 24):
    """The drop-in boundary: the graph as raw SmallEdge records (stream order, flagged-first pairs, duplicates - a stream the
    reference semantics reduce to exactly g), one pinned slab at a time through hb_append_edges, then hb_finalize."""
    total = g.stream_len(salt)
    slab = min(slab, max(total, 1))
    from stract_amd import _lib
    pinned = _lib.PinnedRecords(slab)  # hb_pinned_alloc (hipHostMalloc): what the Rust shim fills batch by batch
    buf = pinned.array
    h2d = ctx.h2d_rate(buf) if hasattr(ctx, "h2d_rate") else None
    s_fill = s_append = 0.0
    at = 0
    while at < total:
        t0 = time.perf_counter()
        k = g.stream_fill(buf, at, salt)
        t1 = time.perf_counter()
        ctx.append_edges(buf[:k])
        s_fill += t1 - t0
        s_append += time.perf_counter() - t1
        at += k
    t0 = time.perf_counter()
    ctx.finalize()
    s_fin = time.perf_counter() - t0
    st = ctx.stats()
    ok = st["n"] == g.n and st["m_eff"] == g.m and st["m_input"] == total
    if not ok:
        raise RuntimeError("ingest of the record stream did not reduce to the clean graph: n %d/%d m_eff %d/%d" % (st["n"], g.n, st["m_eff"], g.m))
    del buf
    pinned.close()
    return {"path": "hb_append_edges x %d + hb_finalize (GPU ingest, device planner)" % ((total + slab - 1) // slab),
            "pinned_h2d_GBs": None if h2d is None else round(h2d, 1),
            "records": total, "record_GB": round(total * 40 / 1e9, 2), "flagged_or_duplicate_records": total - int(g.m),
            "s_fill_slabs_host": round(s_fill, 2), "s_append_edges": round(s_append, 2), "s_finalize": round(s_fin, 2),
            "append_GBs": round(total * 40 / max(s_append, 1e-9) / 1e9, 2),
            "records_per_s": round(total / max(s_append + s_fin, 1e-9)),
            "ingest_peak_device_bytes": int(st["ingest_peak_bytes"]), "ingest_peak_bytes_per_record": round(st["ingest_peak_bytes"] / max(total, 1), 2),
            "allocator_held_peak_bytes": int(st["pool_peak_bytes"]), "m_unique": int(st["m_unique"])}


def end_to_end(a, g, ref_sig, ref_passes, salt=2, seg_records=1 << 24):
    """The drop-in measured end to end: what `stract centrality harmonic <webgraph> <out>` does
    (entrypoint/centrality.rs:41-71) as ONE chain of library calls on the same graph, seconds per stage:
        open the webgraph's edge store + stream it   hb_load_webgraph (CRC-32 of every .col file checked; native column reader,
                                                      pinned double buffer, GPU ingest, device planner)
        HarmonicCentrality::calculate                 hb_run
        the (NodeID, f64) list + harmonic_rank + store_harmonic   hb_store_harmonic_results (both speedy_kv databases; result list, ranks and
                                                      key order from the device)
    Harness (not timed): the record stream is written as an edge store by tests/tantivy_fixture.py (a Python restatement of the
    tantivy serialisers: format unpinned), segment by segment; a sample of keys is read back from the written databases with
    tests/speedy_kv_reader.py."""
    import ctypes
    import shutil
    import tempfile
    from stract_amd import _lib, webgraph
    from tests import speedy_kv_reader as kv
    from tests import tantivy_fixture as tf

    total = g.stream_len(salt)
    store_bytes = total * 40
    base = a.e2e_dir
    if not base:
        disk = os.environ.get("TMPDIR") or "/tmp"
        try:
            free = shutil.disk_usage(disk).free
        except OSError:
            free = 0
        base = disk if free > store_bytes * 1.25 + 64 * (g.n + 1) else "/dev/shm"
    try:
        room = shutil.disk_usage(base).free
    except OSError:
        room = 0
    if room < store_bytes * 1.1 + 64 * (g.n + 1):
        # the HARNESS cannot put its input anywhere: nothing of the product ran, so this is reported, not a failure
        return {"skipped": "no medium with room for the %.1f GB edge store the harness writes (%s has %.1f GB free)" % (store_bytes / 1e9, base, room / 1e9)}
    work = tempfile.mkdtemp(prefix="hb_e2e_", dir=base)
    out = {"medium": "tmpfs (/dev/shm: the store does not fit on the box's disk; read rates are memory rates)" if base.startswith("/dev/shm") else "disk (%s)" % base,
           "records": int(total), "store_GB": round(store_bytes / 1e9, 2)}
    try:
        lib = _lib.load()

        def segment(b):
            def make():
                part = np.empty(min(seg_records, total - b), dtype=_lib.EDGE)
                g.stream_fill(part, b, salt)
                return part
            return make

        t0 = time.perf_counter()
        edges_dir = os.path.join(work, "webgraph", "edges")
        # (4 segments in the making at once: generation, assembly, checksum and write of different files overlap)
        tf.write_edge_store_streamed(edges_dir, (segment(b) for b in range(0, total, seg_records)), workers=4,
                                     crc32=lambda buf: lib.hbw_debug_crc32(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
        os.sync()  # the harness's writes are on the medium before the timed chain starts (no write-back competing with it)
        out["s_harness_write_edge_store"] = round(time.perf_counter() - t0, 2)
        out["segments"] = (total + seg_records - 1) // seg_records
        with _lib.Context() as ctx:
            t0 = time.perf_counter()
            webgraph.load_webgraph(ctx, edges_dir, verify_crc=True)
            t1 = time.perf_counter()
            st = ctx.stats()
            os.environ["HB_TRACE_RESULTS"] = "1"  # (stderr: where the FIRST run of a fresh context spends its time - the timed lines are unaffected)
            run = ctx.run()
            os.environ.pop("HB_TRACE_RESULTS", None)
            t2 = time.perf_counter()
            # [r5] hb_store_harmonic_results: the (NodeID, f64) list, the ranks (device sort), the key order of both databases (device
            # sort) and the files, in one call on the context; rounds 3-4 timed hb_result_copy + hb_result_ranks and the host-sorted
            # hb_store_harmonic apart (C4: 0.9-2.2 s + 6.7 s)
            ctx.store_harmonic(os.path.join(work, "centrality"))
            t4 = time.perf_counter()
            ids, vals = ctx.results()  # (harness: the checks below)
            ranks = ctx.ranks()
        sig = (len(vals), int(vals.view(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF)) if len(vals) else (0, 0)
        stages = {"s_load_webgraph": t1 - t0, "s_run": t2 - t1, "s_store_harmonic": t4 - t2}
        out["s_results_and_ranks"] = "inside s_store_harmonic (hb_store_harmonic_results: results + ranks + device key sort + both databases)"
        tot = t4 - t0
        out.update({k: round(v, 3) for k, v in stages.items()})
        out.update({"s_total": round(tot, 3), "compute_share": round(stages["s_run"] / tot, 4),
                    "load_records_per_s": round(total / stages["s_load_webgraph"]), "load_GBs": round(store_bytes / stages["s_load_webgraph"] / 1e9, 2),
                    "results": int(len(vals)), "store_entries_per_s_per_db": round(len(vals) / max(stages["s_store_harmonic"], 1e-9)),
                    "store_entries_per_s_both_dbs": round(2 * len(vals) / max(stages["s_store_harmonic"], 1e-9)),
                    "ms_ingest_reduce": round(st["ms_ingest"], 1), "ms_plan": round(st["ms_plan"], 1), "ms_state": round(st["ms_h2d"], 1),
                    "ingest_peak_bytes_per_record": round(st["ingest_peak_bytes"] / max(total, 1), 2),
                    "allocator_held_peak_bytes": int(st["pool_peak_bytes"]),
                    "graph_ok": bool(st["n"] == g.n and st["m_eff"] == g.m and st["m_input"] == total),
                    "passes": int(run["passes"]), "same_result_as_record_leg": bool(sig == tuple(ref_sig) and int(run["passes"]) == int(ref_passes))})
        # read a sample back from both databases (harness)
        ok = True
        if len(vals):
            rng = np.random.default_rng(11)
            pick = rng.integers(0, len(vals), 64)
            ints = kv.ids_to_ints(ids[pick])
            db_c = kv.Db(os.path.join(work, "centrality", "harmonic"), "f64", tempfile.gettempdir())  # (the reader builds a small shim .so: not on a noexec tmpfs)
            db_r = kv.Db(os.path.join(work, "centrality", "harmonic_rank"), "u64", tempfile.gettempdir())
            ok = len(db_c) == len(vals) and len(db_r) == len(vals)
            for j, key in zip(pick.tolist(), ints):
                ok = ok and db_c.get(key) == float(vals[j]) and db_r.get(key) == int(ranks[j])
        out["stores_read_back_ok"] = bool(ok)
        out["stores_bytes"] = int(sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(work, "centrality")) for f in fs))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def per_pass_avg(all_pass_stats, n, m_eff, rows_in, work_rows, init_streamed):
    T = len(all_pass_stats[-1]) if all_pass_stats else 0
    avg = []
    for t in range(T):
        rows = [s[t] for s in all_pass_stats if len(s) == T]
        d = dict(rows[-1])
        for k in ("ms_gpu", "ms_main", "ms_level1", "ms_collective"):
            d[k] = float(np.mean([r[k] for r in rows]))
        d["alg_bytes"] = pass_bytes(d, n, m_eff, rows_in)
        d["moved_bytes"] = moved_bytes(d, n, m_eff, rows_in, work_rows, init_streamed)
        avg.append(d)
    return avg


def roofline_of(avg, stats, steps, n, m_eff, init_streamed, config):
    """roofline object of the JSON line: headline = the whole generic dense pass; kernels[] = the launches /
    passes the review tracks, each with its own algorithmic bytes and event-timed duration."""
    dense = [d for d in avg if d["mode"] == 0 and (d["pass"] > 0 or not init_streamed)]
    only_pass0 = False
    if not dense:  # forced bitmap / sweep configurations: pass 0 is the only dense pass; it is booked with what it moves
        dense = [d for d in avg if d["mode"] == 0]
        only_pass0 = init_streamed
    if not dense:
        return None
    rows_in = int(stats["rows_with_in_edges"])
    gbs = lambda b, ms: b / (max(ms, 1e-9) * 1e-3) / 1e9
    b_dense = sum(d["moved_bytes" if only_pass0 else "alg_bytes"] for d in dense)
    ms_dense = sum(d["ms_gpu"] - d["ms_collective"] for d in dense)
    achieved = gbs(b_dense, ms_dense)
    l1_edges, direct = int(stats["level1_edges"]), int(stats["direct_edges"])
    l1_ms = float(np.mean([d["ms_level1"] for d in dense]))
    kernels = []
    dom = None
    pmc = _pmc(config)
    if l1_edges and l1_ms > 0:
        dom_b = 68.0 * l1_edges
        tr = pmc.get("hub_level1_dense", {})
        dom = {"kernel": "hbk::pass_kernel<REAL=false,FUSED=false,STATS=false,UNROLL=4,INIT=false,EPI4=false>, level-1 launch (dense pull over the hub chunks)",
               "alg_bytes_per_launch": dom_b, "alg_bytes_def": "68 B x real edges gathered by the launch (%d)" % l1_edges,
               "avg_launch_ms": round(l1_ms, 4), "launches": len(dense) * steps,
               "achieved": round(gbs(dom_b, l1_ms), 1), "frac": round(gbs(dom_b, l1_ms) / HBM_PEAK_GBS, 4),
               "traffic": tr.get("hbm_bytes_per_dispatch"), "l2_hit_rate": tr.get("l2_hit_rate"), "traffic_source": pmc.get("_source")}
        kernels.append(dict(dom, name="hub_level1_dense"))
    node_ms = float(np.mean([d["ms_main"] for d in dense]))
    if node_ms > 0:
        nb = 68.0 * direct + 192.25 * n
        tr = pmc.get("node_rows_dense", {})
        kernels.append({"name": "node_rows_dense", "kernel": "hbk::pass_kernel<REAL=true,FUSED=true,STATS=false,UNROLL=2,INIT=false,EPI4=true> (node rows: direct gathers + partials, "
                        "merge, changed bits, fused estimator + Kahan)", "alg_bytes_per_launch": nb,
                        "alg_bytes_def": "68 B x direct real edges (%d) + 192.25 B x nodes (%d)" % (direct, n), "avg_launch_ms": round(node_ms, 4),
                        "achieved": round(gbs(nb, node_ms), 1), "frac": round(gbs(nb, node_ms) / HBM_PEAK_GBS, 4),
                        "traffic": tr.get("hbm_bytes_per_dispatch"), "l2_hit_rate": tr.get("l2_hit_rate")})
    for d in avg:
        if d["mode"] == 1:
            kernels.append({"name": "bitmap_pass_t%d" % d["pass"], "kernel": "whole pass, frontier_kernel<...> launches (hub levels, then node rows)",
                            "A_t_pct": round(100.0 * d["active_edges"] / m_eff, 2), "alg_bytes": d["alg_bytes"], "ms": round(d["ms_gpu"], 4),
                            "achieved": round(gbs(d["alg_bytes"], d["ms_gpu"]), 1), "frac": round(gbs(d["alg_bytes"], d["ms_gpu"]) / HBM_PEAK_GBS, 4)})
    sweeps = [d for d in avg if d["mode"] == 2]
    if sweeps:
        d = sweeps[0]
        kernels.append({"name": "first_sweep_pass_t%d" % d["pass"], "kernel": "whole pass, sweep_collect/expand/rows launches",
                        "A_t_pct": round(100.0 * d["active_edges"] / m_eff, 3), "alg_bytes": d["alg_bytes"], "moved_bytes_model": d["moved_bytes"],
                        "ms": round(d["ms_gpu"], 4), "achieved": round(gbs(d["alg_bytes"], d["ms_gpu"]), 1),
                        "frac": round(gbs(d["alg_bytes"], d["ms_gpu"]) / HBM_PEAK_GBS, 4),
                        "frac_moved": round(gbs(d["moved_bytes"], d["ms_gpu"]) / HBM_PEAK_GBS, 4)})
        tail = [x for x in sweeps if x["active_edges"] * 1000 < m_eff]
        if tail:
            kernels.append({"name": "far_tail_passes", "passes": len(tail), "ms_mean": round(float(np.mean([x["ms_gpu"] for x in tail])), 4),
                            "ms_min": round(float(min(x["ms_gpu"] for x in tail)), 4), "bound": "launch latency, not bytes"})
    p0 = avg[0]
    b_loop = sum(d["alg_bytes"] for d in avg)
    mv_loop = sum(d["moved_bytes"] for d in avg)
    ms_loop_gpu = sum(d["ms_gpu"] for d in avg)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": dom["traffic"] if dom else None,
            "what": "whole dense pass (all launches of the pass), algorithmic bytes B_t of SURVEY.md 8(d) over event-timed GPU time, mean over "
                    "the %d dense passes t >= 1%s; `traffic` = PMC HBM bytes of ONE level-1 launch of the dominant kernel" % (
                        len(dense), " (pass 0 apart: it streams 2 B per edge instead of gathering, see pass0)" if init_streamed else ""),
            "frac_of_measured_copy_6.29TBs": round(achieved / HBM_COPY_GBS, 4),
            "pass0": {"alg_bytes": p0["alg_bytes"], "ms": round(p0["ms_gpu"] - p0["ms_collective"], 4),
                      "achieved_on_alg_bytes": round(gbs(p0["alg_bytes"], p0["ms_gpu"] - p0["ms_collective"]), 1),
                      "streamed_initial_registers": bool(init_streamed), "moved_bytes_model": p0["moved_bytes"],
                      "frac_moved": round(gbs(p0["moved_bytes"], p0["ms_gpu"] - p0["ms_collective"]) / HBM_PEAK_GBS, 4)},
            "dominant_kernel": dom, "kernels": kernels,
            "whole_loop": {"alg_bytes": b_loop, "moved_bytes_model": mv_loop, "ms_gpu": round(ms_loop_gpu, 4),
                           "achieved": round(gbs(b_loop, ms_loop_gpu), 1), "frac": round(gbs(b_loop, ms_loop_gpu) / HBM_PEAK_GBS, 4),
                           "frac_moved": round(gbs(mv_loop, ms_loop_gpu) / HBM_PEAK_GBS, 4),
                           "note": "frac books every pass with its 8(d) bytes (pass 0 and the sweep passes move less than that: it flatters them); "
                                   "frac_moved books what this implementation has to move (pass 0: 6 B per edge + 192.25 B per node; sweep passes "
                                   "without the index lists of untouched rows)"},
            "per_pass": [{"t": int(d["pass"]), "mode": int(d["mode"]), "A_t": int(d["active_edges"]),
                          "V_t": (n if d["pass"] == 0 else rows_in if d["mode"] == 0 else int(d["touched"])), "ms": round(d["ms_gpu"], 4),
                          "ms_level1_or_expand": round(d["ms_level1"], 4), "ms_node_rows": round(d["ms_main"], 4), "changed": int(d["changed"]),
                          # frac: on the bytes the pass has to move in this implementation (never > 1); frac_8d: the same time booked
                          # with the SURVEY 8(d) bytes (pass 0 and the sweep passes move far less than those: values > 1 say so)
                          "frac": round(gbs(d["moved_bytes"], d["ms_gpu"]) / HBM_PEAK_GBS, 4),
                          "frac_8d": round(gbs(d["alg_bytes"], d["ms_gpu"]) / HBM_PEAK_GBS, 4)} for d in avg]}


def _die_with_parent():
    """preexec_fn of the child processes: a child must not keep the GPU when this process is killed (PR_SET_PDEATHSIG)."""
    import ctypes
    import signal
    try:
        ctypes.CDLL("libc.so.6").prctl(1, signal.SIGKILL)
    except Exception:
        pass


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and a.gpus > 1:
        # `python bench.py --gpus N` the way the driver invokes `--gpus 1` (no launcher): become the launcher - the same command under
        # torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (VERDICT r5 #3a: the first SCALE run must produce a line)
        if os.environ.get("HB_BENCH_RELAUNCHED") == "1":
            sys.exit("bench.py --gpus %d: relaunched under torch.distributed.run and still alone (WORLD_SIZE=1)" % a.gpus)
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HB_BENCH_RELAUNCHED="1", MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL between processes needs dmabuf IPC on these hosts
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without a launcher: re-executing as `%s`\n" % (a.gpus, " ".join(cmd[1:8]) + " ... bench.py " + " ".join(sys.argv[1:])))
        sys.stderr.flush()
        os.execve(sys.executable, cmd, env)
    if world > 1 and a.gpus > 1 and world != a.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (a.gpus, world))
    # PyTorch is plumbing for N > 1 only (torch.distributed: rendezvous, barrier, the max over ranks).  At N = 1 it is not
    # imported at all: the library then runs on /opt/rocm's HIP runtime instead of the one bundled with torch (same soname:
    # whichever is loaded first serves both), and no second OpenMP runtime enters the process.
    torch, td = None, None
    if world > 1:
        import torch
        import torch.distributed as td

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.collectives == "host-staged":
            td.init_process_group("gloo")
        else:
            if not torch.cuda.is_available():
                sys.exit("bench.py needs a GPU: the HyperBall library has no CPU fallback")
            torch.cuda.set_device(local_rank)
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    staged = world > 1 and a.collectives == "host-staged"

    from stract_amd import _lib, dist, synth
    if hasattr(_lib.load(), "hb_simt_interpreter") and not (staged and os.environ.get("HB_BENCH_FUNCTIONAL") == "1"):
        # (tests/test_simt.py drives the N > 1 CONTROL FLOW of this script on the interpreted test build: functional, never a number)
        raise SystemExit("bench.py measures the gfx950 library only (tests/simt is CPU-side test infrastructure, never a compute path)")
    ndev = _lib.device_count()
    device = local_rank % max(ndev, 1) if staged else local_rank
    on_cuda = torch is not None and not staged

    if world == 1 and _lib.device_count() < 1:
        sys.exit("bench.py needs a GPU: the HyperBall library has no CPU fallback")
    live_ctx = [None]  # the context whose stream the N = 1 barrier synchronises

    def barrier():
        if not on_cuda:
            if live_ctx[0] is not None:
                live_ctx[0].synchronize()
            if td is not None:
                td.barrier()
                if live_ctx[0] is not None:
                    live_ctx[0].synchronize()
            return
        torch.cuda.synchronize()
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    # ---- synthetic input (identical on every rank, deterministic seed)
    t0 = time.perf_counter()
    shared_dir = None
    if world > 1 and a.config in synth.CONFIGS and not os.environ.get("HB_SYNTH_CACHE"):
        # N ranks on one node: local rank 0 generates (C4: ~70 s on all host cores, 11 GB) and leaves the reduced arrays in shared
        # memory, the others map them - not N generators competing for the same cores and N copies of the graph
        shared_dir = "/dev/shm/hb_bench_graph_%s_%s" % (os.environ.get("MASTER_PORT", "0"), a.config)
        os.environ["HB_SYNTH_CACHE"] = shared_dir
        if local_rank == 0:
            import shutil
            shutil.rmtree(shared_dir, ignore_errors=True)
            g, scale, label = synth.make_config(a.config)
        td.barrier()
        if local_rank != 0:
            g, scale, label = synth.make_config(a.config)
        os.environ.pop("HB_SYNTH_CACHE", None)
    else:
        g, scale, label = synth.make_config(a.config)
    t_gen = time.perf_counter() - t0
    n, m_eff = int(g.n), int(g.m)
    tune = tuple(int(x) for x in a.tune.split(",")) if a.tune else ()
    init_streamed = not (a.flags & _lib.HB_FLAG_NO_INIT_PASS)

    def run_leg(partition, changed_only=False):
        """One decomposition: context, load, W + K runs.  Returns (ctx, measurement, load info)."""
        flags = a.flags
        rccl_id = None
        if world > 1:
            if staged:
                flags |= _lib.HB_FLAG_NO_RCCL
            else:
                rccl_id = dist.torch_unique_id(rank, world)
            if partition == "dest":
                flags |= _lib.HB_FLAG_DEST_PARTITION
            if changed_only:
                flags |= _lib.HB_FLAG_CHANGED_ONLY
        ctx = _lib.Context(device=device, flags=flags, chunk=a.chunk, rank=rank, world_size=world, rccl_id=rccl_id, tune=tune)
        if staged:
            ctx._staged_collectives = dist.HostStagedCollectives(ctx)  # (kept alive with the context: it owns the callbacks)
        t0 = time.perf_counter()
        info = {"path": "hb_load_dense (bench-only export: pre-reduced CSR)"}
        if world > 1:
            split = dist.partition_dense_by_dest if partition == "dest" else dist.partition_dense
            rp, src = split(g.row_ptr, g.src, rank, world)
            ctx.load_dense(g.ids, rp, src)
        elif a.input == "records":
            info = load_records(ctx, g)  # raises if the stream does not reduce to the clean graph: the bench fails, loudly
        else:
            ctx.load_dense(g.ids, g.row_ptr, g.src)
        info["s_load"] = round(time.perf_counter() - t0, 2)
        live_ctx[0] = ctx
        return ctx, measure(ctx, a.steps, a.warmup, barrier, td, torch), info

    exit_code = 0
    main_part = "single" if world == 1 else ("dest" if a.partition == "dest" else "edge")
    ctx, ms, load = run_leg(main_part, a.changed_only and world > 1)
    ids, vals = ctx.results()
    stats = ms["stats"]
    passes = ms["passes"]
    last_pass_stats = ms["pass_stats"][-1] if ms["pass_stats"] else []

    # ---- parity + CPU baseline (rank 0 computes; a checksum rerun, if needed, is collective)
    cpu, parity = None, None
    cores_used = [None]
    if a.cpu_seconds > 0 or a.verify:
        cpu, parity = cpu_and_parity(a, g, ctx, td, rank, world, passes, ids, vals, last_pass_stats, cores_out=cores_used)

    steps = max(a.steps, 1)
    out = None
    if rank == 0:
        teps = m_eff * passes * steps / ms["dt"]
        rows_in = int(stats["rows_with_in_edges"])
        avg = per_pass_avg(ms["pass_stats"], n, m_eff, rows_in, int(stats["work_rows"]), init_streamed)
        roof = roofline_of(avg, stats, steps, n, m_eff, init_streamed, a.config)
        gathered = sum(min(int(d["active_edges"]), m_eff) if d["pass"] else m_eff for d in avg)
        loop_ms = ms["loop_ms"]
        out = {
            "metric": "HyperBall traversed edges/sec (GTEPS)",
            "value": round(teps / 1e9, 4),
            "unit": "GTEPS",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms["dt"] * 1e3 / steps, 3),
            # wall time of the FIRST hb_run of the freshly loaded context (= what a drop-in user sees: calculate() runs once) and its ratio to
            # the steady-state step; null when --warmup 0 (then the first run is the first timed step)
            "first_run_ms": None if "first_run_ms" not in ms else round(ms["first_run_ms"], 3),
            "first_run_over_steady": None if "first_run_ms" not in ms else round(ms["first_run_ms"] / (ms["dt"] * 1e3 / steps), 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if not staged else "synthetic; FUNCTIONAL RUN (--collectives host-staged: gloo + host-staged exchanges, %d device(s) for %d ranks): not a measurement" % (max(ndev, 1), world),
            "config": {"workload": "%s %s" % (a.config, label),
                       "n_hosts": n, "m_eff": m_eff, "passes_T": passes,
                       "parallelism": ("1 GPU" if world == 1 else
                                       "destination-partition x%d + allgather(u8)/pass" % world if main_part == "dest" else
                                       "edge-partition x%d + allreduce(max,u8)/pass" % world)},
            "parity_bit_exact": None if parity is None else parity["bit_exact"],
            "parity": parity,
            "roofline": roof,
            "cpu_baseline": cpu,
            "detail": {"input": load,
                       "ms_loop_per_step": round(loop_ms / steps, 3), "ms_gpu_passes_per_step": round(ms["gpu_ms"] / steps, 3),
                       "ms_collective_per_step": round(ms["coll_ms"] / steps, 3), "ms_finish_per_step": round(ms["d2h_ms"] / steps, 3),
                       "ms_finish_first_run": None if "first_run_finish_ms" not in ms else round(ms["first_run_finish_ms"], 3),
                       "loop_gteps": round(m_eff * passes / (loop_ms / steps * 1e-3) / 1e9, 4) if loop_ms else None,
                       "gathered_edges_per_run": gathered,
                       "gathered_gteps": round(gathered / (loop_ms / steps * 1e-3) / 1e9, 4) if loop_ms else None,
                       "collective": wire_info(world, main_part, a.changed_only, stats, n, ms, steps) if world > 1 else None,
                       "scaling_model": scaling_model(avg, n, m_eff, passes, ms["dt"] * 1e3 / steps) if world == 1 and avg else None,
                       "results": int(len(vals)), "s_generate": round(t_gen, 2), "s_load": load["s_load"],
                       "ms_plan": round(stats["ms_plan"], 1), "ms_h2d": round(stats["ms_h2d"], 1),
                       "device_bytes": int(stats["device_bytes"]), "virtual_rows": int(stats["virtual_rows"]),
                       "level1_edges": int(stats["level1_edges"]), "direct_edges": int(stats["direct_edges"])},
        }
        if a.pass_log:
            with open(a.pass_log, "w") as f:
                json.dump({"config": out["config"], "passes": avg}, f, indent=1)
    ref_sig = (len(vals), int(vals.view(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF)) if len(vals) else (0, 0)
    ctx.close()

    # ---- N > 1: the other decompositions, same graph, same K / W (extra legs; `value` stays the edge partition)
    if world > 1 and a.partition == "both":
        # The main line must survive the extra legs: a rank that fails inside one leaves the others waiting in a collective,
        # and a blocked ctypes call cannot be interrupted from Python.  A watchdog THREAD prints the line as it stands and ends
        # every rank once the legs exceed their allowance.
        import threading
        legs = {}

        def give_up():
            if rank == 0:
                legs.setdefault("error", "extra legs exceeded %d s and were abandoned; `value` (edge partition) is unaffected" % a.legs_timeout)
                out["detail"]["partitions"] = legs
                print(json.dumps(out), flush=True)
            sys.stderr.write("bench.py rank %d: extra partition legs timed out\n" % rank)
            sys.stderr.flush()
            if shared_dir and local_rank == 0:
                import shutil
                shutil.rmtree(shared_dir, ignore_errors=True)
            os._exit(0)

        dog = threading.Timer(a.legs_timeout, give_up)
        dog.daemon = True
        dog.start()
        for name, part, co in (("edge_changed_only", "edge", True), ("dest_allgather", "dest", False), ("dest_changed_only", "dest", True)):
            try:
                c2, m2, _ = run_leg(part, co)
                i2, v2 = c2.results()
                same = (len(v2), int(v2.view(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF) if len(v2) else 0) == ref_sig
                if rank == 0:
                    legs[name] = {"value": round(m_eff * m2["passes"] * steps / m2["dt"] / 1e9, 4), "unit": "GTEPS",
                                  "ms_per_step": round(m2["dt"] * 1e3 / steps, 3), "same_result_as_edge_partition": bool(same),
                                  "collective": wire_info(world, part, co, m2["stats"], n, m2, steps)}
                c2.close()
            except Exception as e:  # the other ranks are now alone in a collective: the watchdog ends the run
                legs[name] = {"error": str(e)[:300]}
                sys.stderr.write("bench.py rank %d: leg %s failed: %s\n" % (rank, name, e))
                break
        else:
            dog.cancel()
        if rank == 0:
            out["detail"]["partitions"] = legs
            # `value` stays the decomposition BASELINE configs[3] names (edge partition + all-reduce per pass); the fastest decomposition
            # whose final results equal the main leg's bit for bit is named at the top level, where a truncating reader still sees it
            ok = {k: v for k, v in legs.items() if isinstance(v, dict) and v.get("same_result_as_edge_partition") and "value" in v}
            ok["edge_allreduce"] = {"value": out["value"], "ms_per_step": out["ms_per_step"]}
            best = max(ok, key=lambda k: ok[k]["value"])
            out["best_decomposition"] = {"name": best, "value": ok[best]["value"], "unit": "GTEPS", "ms_per_step": ok[best]["ms_per_step"],
                                         "note": "same final (NodeID, f64) list as the edge-partition leg; `value` above is the north-star decomposition"}
        if dog.is_alive() and any("error" in v for v in legs.values() if isinstance(v, dict)):
            dog.join()  # a leg failed here: wait for the watchdog (it prints on rank 0 and ends the process)

    # ---- the whole reference command on the same graph: store -> load -> run -> ranks -> store_harmonic
    want_e2e = world == 1 and rank == 0 and (a.end_to_end == "on" or (a.end_to_end == "auto" and a.input == "records" and a.config in ("C3", "C4", "LT")
                                                          and not a.flags and not a.tune and not a.chunk))
    if want_e2e:
        out["detail"]["end_to_end"] = end_to_end(a, g, ref_sig, passes)
        e2e = out["detail"]["end_to_end"]
        if "skipped" not in e2e and not (e2e["graph_ok"] and e2e["same_result_as_record_leg"] and e2e["stores_read_back_ok"]):
            exit_code = 4  # the chain produced something else than the record leg: loud

    # ---- the other BASELINE config as an extra leg (1 GPU, child process): C3 beside the default C4 line (C4 beside another main
    # config with --c4-leg on)
    plain = not a.flags and not a.tune and not a.chunk
    legs_wanted = []
    if world == 1 and a.config != "C3" and (a.c3_leg == "on" or (a.c3_leg == "auto" and a.config == "C4" and plain)):
        legs_wanted.append("C3")
    if world == 1 and a.config != "C4" and a.c4_leg == "on":
        legs_wanted.append("C4")
    if legs_wanted:
        g.close()
        del g
        live_ctx[0] = None
        for cfg in legs_wanted:
            try:
                leg = sub_leg(a, cfg)
            except Exception as e:  # the main line is still printed, but the run counts as failed (exit code below)
                leg = {"error": str(e)[:400]}
            out["detail"][cfg.lower()] = leg
            if "error" in leg:
                exit_code = 3
    if rank == 0:
        print(json.dumps(out), flush=True)
    if td is not None:
        td.barrier()
        if shared_dir and local_rank == 0:
            import shutil
            shutil.rmtree(shared_dir, ignore_errors=True)
        td.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


XGMI_LINK_GBS_PER_DIR = 76.8  # MI355X: 7 xGMI links per GPU, ~153.6 GB/s each counting both directions (one direct link per peer)


def scaling_model(avg, n, m_eff, passes, ms_per_step):
    """What DESIGN.md §6's cost model predicts for N = 2, 4, 8 from THIS run's per-pass numbers, per decomposition - written into the
    N = 1 line so that BENCH and SCALE records can be read side by side (VERDICT r5 #3c).  A model, not a measurement: local compute of a
    pass = its 1-GPU time / N (optimistic: the node rows of the edge partition do not shrink with N), every GPU sends on its N - 1
    direct xGMI links at once at 76.8 GB/s per link and direction, no protocol overhead, no launch latency.  Bytes received per GPU and
    pass: edge partition 2 (N-1)/N n 64 (all-reduce = reduce-scatter + all-gather); its changed-only form the same over the rows that
    changed + the ranks' n/8-byte bitmaps; destination partition (N-1)/N (n 64 + n/8); its changed-only form (N-1)/N (changed 64 + n/8),
    and with the 6-bit register packing of the packed exchange 48 instead of 64 bytes per changed row."""
    n_pad = (n + 63) // 64 * 64
    fixed_ms = max(ms_per_step - sum(d["ms_gpu"] for d in avg), 0.0)  # hb_begin, hb_finish, host gaps: not divided
    out = {"assumptions": "local compute per pass = 1-GPU pass time / N; N-1 links x %.1f GB/s per direction; bytes at link rate, no overlap / full overlap of "
                          "collective and compute; fixed %.2f ms per run outside the passes" % (XGMI_LINK_GBS_PER_DIR, fixed_ms), "per_n": {}}
    for N in (2, 4, 8):
        egress = (N - 1) * XGMI_LINK_GBS_PER_DIR  # GB/s
        f = (N - 1) / N
        legs = {}
        for name in ("edge_allreduce", "edge_changed_only", "dest_allgather", "dest_changed_only", "dest_changed_only_6bit"):
            t_sum = t_max = 0.0
            for d in avg:
                ch = min(int(d["changed"]), n)
                if name == "edge_allreduce":
                    b = 2 * f * n_pad * 64
                elif name == "edge_changed_only":
                    b = 2 * f * ch * 64 + (N - 1) * n_pad / 8
                elif name == "dest_allgather":
                    b = f * (n_pad * 64 + n_pad / 8)
                elif name == "dest_changed_only":
                    b = f * (ch * 64 + n_pad / 8)
                else:
                    b = f * (ch * 48 + n_pad / 8)
                coll = b / egress / 1e6
                local = d["ms_gpu"] / N
                t_sum += local + coll
                t_max += max(local, coll)
            legs[name] = {"ms_per_run_no_overlap": round(t_sum + fixed_ms, 2), "ms_per_run_full_overlap": round(t_max + fixed_ms, 2),
                          "gteps_no_overlap": round(m_eff * passes / ((t_sum + fixed_ms) * 1e-3) / 1e9, 1),
                          "gteps_full_overlap": round(m_eff * passes / ((t_max + fixed_ms) * 1e-3) / 1e9, 1)}
        out["per_n"][str(N)] = legs
    return out


def wire_info(world, part, changed_only, stats, n, ms, steps):
    """What travelled, and what the cost model of DESIGN.md §6 predicts for it - so that the first record measured on N > 1 physical
    GPUs can be read against a number written down BEFORE it: every GPU has one direct link to each peer, so with N ranks it can send
    on N - 1 links at once; an all-reduce is a reduce-scatter + an all-gather (each moves (N-1)/N of the buffer out of every GPU),
    an all-gather moves every other rank's slice in.  Lower bounds: no protocol overhead, no launch latency."""
    n_pad = (n + 63) // 64 * 64
    passes = max(ms["passes"], 1)
    egress = max(world - 1, 1) * XGMI_LINK_GBS_PER_DIR  # GB/s out of (and into) one GPU
    ar_bytes = 2.0 * (world - 1) / world * n_pad * 64
    ag_bytes = (world - 1) / world * (n_pad * 64 + n_pad / 8)
    ar_ms = ar_bytes / 2.0 / egress / 1e6  # the two halves each move (N-1)/N * S out of every GPU, link-parallel
    ag_ms = ag_bytes / egress / 1e6
    measured = ms["coll_ms"] / steps / passes
    local = (ms["gpu_ms"] - ms["coll_ms"]) / steps / passes
    pred = 2 * ar_ms if part == "edge" else ag_ms
    if changed_only and stats["wire_bytes"]:
        pred = float(stats["wire_bytes"]) / passes / egress / 1e6  # what this run really received, at link rate
    return {"ran": part + ("+changed-only" if changed_only else ""),
            "received_bytes_per_gpu_per_run": int(stats["wire_bytes"]),
            "edge_allreduce_bytes_per_gpu_per_pass": ar_bytes,
            "dest_allgather_bytes_per_gpu_per_pass": ag_bytes,
            "ms_collective_per_pass": round(measured, 4),
            "model": {"xgmi_GBs_per_link_per_direction": XGMI_LINK_GBS_PER_DIR, "links_used_per_gpu": max(world - 1, 1),
                      "edge_allreduce_ms_per_pass_lower_bound": round(2 * ar_ms, 3), "dest_allgather_ms_per_pass_lower_bound": round(ag_ms, 3),
                      "this_leg_ms_collective_per_pass_lower_bound": round(pred, 3),
                      "measured_ms_local_compute_per_pass": round(local, 4),
                      "predicted_ms_per_pass_no_overlap": round(local + pred, 3), "predicted_ms_per_pass_full_overlap": round(max(local, pred), 3),
                      "note": "lower bounds at link rate; measured / predicted > 1 is protocol + launch overhead, < 1 means the model is wrong"}}


def sub_leg(a, config):
    """Another BASELINE config on one GPU, appended to the default line so that it is measured under the driver's clock too:
    GTEPS, whole-dense-pass and dominant-kernel fractions, parity (final list), its own end-to-end chain.  Round 5: the default line
    is configs[3] (C4, the north-star graph) and this leg is configs[2] (C3, the headline of rounds 1-4: 5 warm-up + 20 timed
    runs); `--c4-leg on` appends C4 (1 + 2 runs) to another main config.  Runs as a CHILD process (this script with --config ...):
    the main process has just released a graph of tens of GB, and a child that is killed must not take the main line down with it."""
    runs = ["--steps", "20", "--warmup", "5"] if config == "C3" else ["--steps", "2", "--warmup", "1"]
    cmd = [sys.executable, os.path.abspath(__file__), "--config", config] + runs + ["--c4-leg", "off", "--c3-leg", "off", "--parity", a.parity,
           "--cpu-seconds", str(a.cpu_seconds), "--input", a.input, "--end-to-end", a.end_to_end] + (["--verify"] if a.verify else []) + (
               ["--e2e-dir", a.e2e_dir] if a.e2e_dir else [])
    env = dict(os.environ, HB_BENCH_CHILD="1")  # measured in the child itself, not under another supervisor
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3000, env=env, preexec_fn=_die_with_parent)
    if r.stderr:
        sys.stderr.write("---- stderr of the %s leg ----\n" % config + r.stderr[-20000:] + "\n---- end of the %s leg ----\n" % config)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        return {"error": "child exited with %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
    d = json.loads(line)
    roof = d.get("roofline") or {}
    keep = ("frac", "achieved", "frac_of_measured_copy_6.29TBs", "pass0", "dominant_kernel", "kernels", "whole_loop", "per_pass")
    det = d.get("detail", {})
    return {"workload": d["config"]["workload"], "n_hosts": d["config"]["n_hosts"], "m_eff": d["config"]["m_eff"], "passes_T": d["config"]["passes_T"],
            "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
            "first_run_ms": d.get("first_run_ms"), "first_run_over_steady": d.get("first_run_over_steady"),
            "parity_bit_exact": d["parity_bit_exact"], "parity": d["parity"], "roofline": {k: roof[k] for k in keep if k in roof},
            "cpu_baseline": d["cpu_baseline"], "input": det.get("input"), "end_to_end": det.get("end_to_end"), "s_generate": det.get("s_generate"),
            "device_bytes": det.get("device_bytes"), "results": det.get("results")}


def _pmc(config):
    """PMC numbers per launch class: NOT measured by this run - read from the newest committed rocprofv3 --pmc summary of the same
    command (profiles/current_<config>_pmc.json, written by tools/profile.sh + tools/export_profile.py, which splits the dispatches
    of one kernel template by launch: level 1 / level 2 ...).  {} when there is none."""
    rel = os.path.join("profiles", "current_%s_pmc.json" % config)
    try:
        with open(os.path.join(ROOT, rel)) as f:
            d = json.load(f)
    except Exception:
        return {}
    out = {"_source": rel + " (committed rocprofv3 --pmc runs of `bench.py --config %s`, per launch class)" % config}
    for k, v in d.items():
        if not isinstance(v, dict):
            continue
        if not k.startswith("hbk::pass_kernel<"):
            continue
        # template arguments: <REAL, FUSED, STATS, UNROLL, INIT, EPI4>; "#L<level>" = launch class of the hub-chunk kernel
        targs = [x.strip() for x in k[len("hbk::pass_kernel<"):k.index(">")].split(",")]
        if len(targs) != 6 or targs[2] != "false" or targs[4] != "false":
            continue  # pass-statistics builds and the pass-0 (INIT) instantiations are other launches
        if targs[0] == "false" and k.endswith("#L1"):
            out["hub_level1_dense"] = dict(v, _kernel=k)
        if targs[0] == "true" and targs[1] == "true":
            out["node_rows_dense"] = dict(v, _kernel=k)
    return out


def _cpu_quota():
    """CPUs the cgroup lets this process use (cgroup v2 cpu.max or v1 cfs quota), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(round(float(q) / float(p))))
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, int(round(q / p)))
    except Exception:
        pass
    return None


def _pick_threads(hbo, synth, ncpu, quota):
    """OpenMP thread count that is fastest on THIS box (SMT, NUMA, container CPU quotas): two dense passes
    of a small calibration graph per candidate."""
    cal = synth.RmatGraph(19, 4_000_000)
    best_t, cores = None, ncpu
    cands = {ncpu, max(ncpu // 2, 1), max(ncpu // 4, 1), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}
    if quota:
        cands |= {min(ncpu, quota), min(ncpu, 2 * quota), min(ncpu, 4 * quota)}
    for th in sorted(cands, reverse=True):
        oc = hbo.Dense(cal.id_low64(), cal.row_ptr, cal.src, threads=th)
        oc.step(0)
        t0 = time.perf_counter()
        oc.step(0)
        oc.step(0)
        dt = time.perf_counter() - t0
        oc.close()
        if best_t is None or dt < best_t:
            best_t, cores = dt, th
    return cores


def cpu_and_parity(a, g, ctx, td, rank, world, gpu_passes, gpu_ids, gpu_vals, gpu_pass_stats, cores=None, faithful=True, cores_out=None):
    """Rank 0: oracle (dense OpenMP port of the reference arithmetic) on this box's host cores, on the
    same graph, for as many passes as fit in --cpu-seconds (all of them with --verify).  Parity: the final
    (NodeID, f64) list when the oracle converged, else a checksum of registers (+ Kahan state on one GPU)
    after the last pass the oracle finished - the GPU is re-run for that many passes (collectively for
    N > 1).  Returns (cpu_baseline dict or None, parity dict)."""
    cpu, parity, done, o = None, None, 0, None
    if rank == 0:
        from oracle import hbo
        from stract_amd import synth

        ncpu = os.cpu_count() or 1
        quota = _cpu_quota()  # containers: the cgroup CPU quota can be far below the visible hardware threads
        if cores is None:
            cores = _pick_threads(hbo, synth, ncpu, quota)
        if cores_out is not None:
            cores_out[0] = cores
        o = hbo.Dense(g.id_low64(), g.row_ptr, g.src, threads=cores)
        full = a.verify or a.parity == "full" or (a.parity == "auto" and world == 1)
        t0 = time.perf_counter()
        has = True
        cpu_pass_s = []
        sample_passes, sample_s = 0, 0.0  # the CPU-baseline sample: the passes that START inside the budget
        while has and (full or time.perf_counter() - t0 < a.cpu_seconds):
            t1 = time.perf_counter()
            in_budget = a.cpu_seconds <= 0 or t1 - t0 < a.cpu_seconds
            has, _ = o.step(hbo.FRONTIER)
            cpu_pass_s.append(time.perf_counter() - t1)
            done += 1
            if in_budget or sample_passes == 0:
                sample_passes, sample_s = done, time.perf_counter() - t0
        dt_all = time.perf_counter() - t0
        dt = sample_s if sample_passes else dt_all
        gpu_same_ms = sum(ps["ms_gpu"] for ps in gpu_pass_stats[:sample_passes])
        cpu = {"value": round(g.m * sample_passes / max(dt, 1e-9) / 1e9, 5), "unit": "GTEPS", "cores": cores, "kind": "port",
               "sample": "first %d of %d passes of the same graph, oracle dense OpenMP port (oracle/hb_oracle.c), %.1f s, "
                         "%d OpenMP threads (fastest of a calibration sweep; %d hardware threads visible, cgroup CPU quota %s)%s"
                         % (sample_passes, gpu_passes, dt, cores, ncpu, quota if quota else "none",
                            "; the oracle then ran on to convergence for the parity check (%d passes, %.1f s in all: not part of the sample)" % (done, dt_all)
                            if done > sample_passes else ""),
               "converged": not has, "seconds": round(dt, 3),
               "gpu_same_passes": {"passes": sample_passes, "ms": round(gpu_same_ms, 3),
                                   "gteps": round(g.m * sample_passes / (gpu_same_ms * 1e-3) / 1e9, 3) if gpu_same_ms else None,
                                   "speedup": round(dt * 1e3 / gpu_same_ms, 1) if gpu_same_ms else None}}
        if quota:
            cpu["cpu_quota"] = quota
        if not has:
            cpu["seconds_to_convergence"] = round(dt_all, 3)
            ovals, keep, k = o.finish()
            same = (done == gpu_passes and k == len(gpu_vals) and np.array_equal(gpu_ids, g.ids[keep]) and
                    np.array_equal(gpu_vals.view(np.uint64), ovals[keep].view(np.uint64)))
            parity = {"bit_exact": bool(same), "scope": "final (NodeID, f64) list after all %d passes, %d results" % (done, k),
                      "oracle": "oracle/hb_oracle.c dense form (parity unpinned against the Rust reference, see DESIGN.md)"}
        # the structure-faithful single-thread form (what `stract centrality harmonic` does), C1 only
        try:
            if not faithful:
                raise LookupError
            c1 = synth.RmatGraph(synth.CONFIGS["C1"]["scale"], synth.CONFIGS["C1"]["m"])
            _, _, fst = hbo.faithful_run(c1.edges())
            cpu["cpu_faithful"] = {"value": round(fst["m_eff"] * fst["passes"] / fst["seconds_loop"] / 1e9, 6), "unit": "GTEPS",
                                   "cores": 1, "sample": "C1 (%d hosts / %d edges), all %d passes, %.2f s, single thread, "
                                   "ordered map + per-pass clone + re-dedup + bloom (hbo.faithful_run)"
                                   % (fst["n"], fst["m_eff"], fst["passes"], fst["seconds_loop"])}
        except LookupError:
            pass
        except Exception as e:  # pragma: no cover
            cpu["cpu_faithful"] = {"error": str(e)}
    # did the oracle stop early?  then compare state checksums after `done` passes
    k = done if (rank == 0 and parity is None) else 0
    if td is not None:
        import torch
        need = torch.tensor([k], dtype=torch.int64, device="cuda" if td.get_backend() == "nccl" else "cpu")
        td.broadcast(need, src=0)
        k = int(need.item())
    if k > 0:
        ctx.begin()
        for _ in range(k):
            ctx.step()
        hr, hk = ctx.state_hash()
        if rank == 0:
            ohr, ohk = o.state_hash()
            same = (hr == ohr) and (world > 1 or hk == ohk)
            parity = {"bit_exact": bool(same),
                      "scope": "checksum of all %d x 64 registers%s after pass %d of %d (the CPU budget ended there)"
                               % (g.n, "" if world > 1 else " and of every Kahan (sum, err)", k, gpu_passes),
                      "oracle": "oracle/hb_oracle.c dense form (parity unpinned against the Rust reference, see DESIGN.md)"}
        ctx.run()  # leave the context finished (results valid)
    if rank == 0 and o is not None:
        o.close()
    if world > 1:
        cpu = None  # the CPU baseline is reported at N = 1 only; parity is reported at every N
    return cpu, parity


if __name__ == "__main__":
    main()
