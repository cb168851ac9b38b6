"""Parity tests proper: the HIP path (through the C ABI, stract_amd._lib) against the CPU
oracle on the same inputs - bit-exact registers, Kahan state, pass counts and final
(NodeID, f64) lists.  Mirrors the reference's own tests for this path
(crates/core/src/webgraph/centrality/harmonic.rs:358-578) plus per-pass state checks the
reference cannot express.  All tests need a gfx950 device; there is no CPU fallback."""
import json
import os

import numpy as np
import pytest

from oracle import hbo
from stract_amd import _lib, dist, synth
from stract_amd.harmonic import EdgeListGraph, HarmonicCentrality
from tests import graphs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hyperball_golden.json")


def _as_map(ids, vals):
    return {str((int(h) << 64) | int(l)): float(v).hex() for l, h, v in zip(ids["lo"], ids["hi"], vals)}


def _oracle_dense(ids, row_ptr, src):
    o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
    T = o.run()
    vals, keep, k = o.finish()
    return o, T, vals, keep, k


def _check_final(ctx, ids, T, vals, keep, st):
    gids, gvals = ctx.results()
    assert st["passes"] == T
    assert np.array_equal(gids, ids[keep])
    assert np.array_equal(gvals.view(np.uint64), vals[keep].view(np.uint64))


# ---- estimator ------------------------------------------------------------------------------
def test_device_estimator_matches_oracle(gpu_ctx_factory):
    gold = json.load(open(GOLD))
    regs = np.array(gold["size_cases"]["registers"], dtype=np.uint8)
    rng = np.random.default_rng(7)
    more = graphs.random_registers(rng, 20_000)
    # counters as they occur in a run: k random items added
    real = np.zeros((3000, 64), dtype=np.uint8)
    for i in range(len(real)):
        for x in rng.integers(0, 1 << 63, size=int(rng.integers(1, 400)), dtype=np.uint64):
            hbo.hll_add(real[i], int(x))
    with gpu_ctx_factory() as ctx:
        assert ctx.hll_size(regs).tolist() == gold["size_cases"]["sizes"]
        for block in (more, real, np.zeros((1, 64), np.uint8), np.full((3, 64), 65, np.uint8)):
            assert np.array_equal(ctx.hll_size(block), hbo.hll_sizes(block))


# ---- the reference's behavioural tests, through the operator mirror ---------------------------
def test_harmonic_centrality_fixture(gpu_ctx_factory):
    # harmonic.rs:460-474
    hc = HarmonicCentrality.calculate(graphs.fixture_graph())
    A, B, C, D = graphs.A, graphs.B, graphs.C, graphs.D
    assert hc.get(C) > hc.get(A) > hc.get(B)
    assert hc.get(D) is None
    assert hc.len() == 3 and not hc.is_empty()
    assert [k for k, _ in hc.iter()] == [A, B, C]
    assert np.float64(hc.get(A)).view(np.uint64) == 0x3FE5555555555555
    assert np.float64(hc.get(B)).view(np.uint64) == 0x3FE38E38E38E38E3
    assert np.float64(hc.get(C)).view(np.uint64) == 0x3FF0000000000000
    assert hc.stats["passes"] == 4


def test_host_harmonic_centrality(gpu_ctx_factory):
    # harmonic.rs:358-458
    g, (a, b, c, d) = graphs.host_fixture()
    hc = HarmonicCentrality.calculate(g)
    assert hc.get(b) > (hc.get(a) or 0.0)
    assert hc.get(a) is None


def test_additional_edges_ignored(gpu_ctx_factory):
    # harmonic.rs:476-528
    base = HarmonicCentrality.calculate(graphs.fixture_graph())
    extra = HarmonicCentrality.calculate(graphs.fixture_graph(extra=[(graphs.A, graphs.B, 0)] * 8))
    assert _as_map(*base.arrays()) == _as_map(*extra.arrays())


@pytest.mark.parametrize("flag", [graphs.TAG, graphs.SAME_ICANN_DOMAIN])
def test_rel_flags_ignored(gpu_ctx_factory, flag):
    # harmonic.rs:530-578
    hc = HarmonicCentrality.calculate(graphs.fixture_graph(flags=flag))
    assert hc.is_empty() and hc.len() == 0
    assert hc.stats["n"] == 4 and hc.stats["m_eff"] == 0


def test_first_occurrence_flag_wins(gpu_ctx_factory):
    A, B, C = 10, 20, 30
    lost = HarmonicCentrality.calculate(EdgeListGraph.from_tuples([(A, B, graphs.NOFOLLOW), (A, B, 0), (B, C, 0)]))
    assert lost.get(B) is None and lost.get(C) is not None
    kept = HarmonicCentrality.calculate(EdgeListGraph.from_tuples([(A, B, 0), (A, B, graphs.NOFOLLOW), (B, C, 0)]))
    assert kept.get(B) is not None and kept.get(C) is not None


def test_empty_and_singleton_graphs(gpu_ctx_factory):
    # n = 0 and n = 1 are defined as "empty result"
    hc = HarmonicCentrality.calculate(EdgeListGraph(np.zeros(0, dtype=_lib.EDGE)))
    assert hc.is_empty() and hc.stats["n"] == 0
    hc = HarmonicCentrality.calculate(EdgeListGraph.from_tuples([(5, 5)]))
    assert hc.is_empty() and hc.stats["n"] == 1
    with gpu_ctx_factory() as ctx:
        with pytest.raises(_lib.HyperballError):
            ctx.run()  # nothing loaded
        with pytest.raises(_lib.HyperballError):
            ctx.results()


def test_release_cached_memory(gpu_ctx_factory):
    """hb_release_cached_memory [ABI 5, ADVICE r4]: what the caching allocator keeps after a context is gone goes back to the runtime on
    request (other allocators in the process cannot reclaim it themselves); contexts that are alive keep what they hold."""
    import ctypes
    lib = _lib.load()
    g = synth.RmatGraph(13, 60_000)
    with gpu_ctx_factory() as ctx:
        ctx.load_edges(g.edges(salt=1, salt_seed=5))   # the ingest's and the planner's work memory is cached after this
        held = ctypes.c_uint64(0)
        assert lib.hb_release_cached_memory(ctypes.byref(held)) == _lib.HB_OK
        st = ctx.run()                                  # the context's own buffers were not touched
        assert st["passes"] > 0
    assert lib.hb_release_cached_memory(ctypes.byref(held)) == _lib.HB_OK
    again = ctypes.c_uint64(1)
    assert lib.hb_release_cached_memory(ctypes.byref(again)) == _lib.HB_OK and again.value == 0  # nothing left to give back
    assert lib.hb_release_cached_memory(None) == _lib.HB_OK


def test_error_behaviour_of_the_boundary(gpu_ctx_factory):
    """Call-order and argument errors come back as negative codes with a message (include/hyperball.h: nothing unwinds, nothing
    computes on bad input) and leave the context usable."""
    lib = _lib.load()
    g = synth.RmatGraph(10, 5000)
    e = g.edges(salt=0)
    with gpu_ctx_factory() as ctx:
        assert lib.hb_append_edges(ctx.h, None, 5) == _lib.HB_ERR_INVALID        # NULL records with a count
        assert lib.hb_append_edges(ctx.h, None, 0) == _lib.HB_OK                 # an empty batch is fine
        for call in (lambda: ctx.begin(), lambda: ctx.step(), lambda: ctx.finish(), lambda: ctx.run()):
            with pytest.raises(_lib.HyperballError):
                call()                                                           # nothing loaded yet
        with pytest.raises(_lib.HyperballError):
            ctx.tail_segment_end()                                               # not a reference-tail context
        with pytest.raises(_lib.HyperballError):
            ctx.append_tail_edges(e[:3])
        with pytest.raises(_lib.HyperballError):
            ctx.load_dense(g.ids[::-1], g.row_ptr, g.src)                        # ids not ascending
        bad_rp = g.row_ptr.copy()
        bad_rp[-1] -= 1
        with pytest.raises(_lib.HyperballError):
            ctx.load_dense(g.ids, bad_rp, g.src)                                 # row_ptr[n] != m
        assert b"" != (lib.hb_last_error(ctx.h) or b"")
        ctx.load_edges(e)                                                        # ... and the context still works
        with pytest.raises(_lib.HyperballError):
            ctx.step_finish()                                                    # no hb_step_local before
        st = ctx.run()
        ids, vals = ctx.results()
        o, T, ov, keep, k = _oracle_dense(g.ids, g.row_ptr, g.src)
        assert st["passes"] == T and len(vals) == k and np.array_equal(vals.view(np.uint64), ov[keep].view(np.uint64))
        small = np.zeros(1, dtype=np.uint64)
        assert lib.hb_result_ranks(ctx.h, small.ctypes.data, 1) < 0              # result buffer too small
    with gpu_ctx_factory(flags=_lib.HB_FLAG_REFERENCE_TAIL) as ctx:
        with pytest.raises(_lib.HyperballError):
            ctx.append_tail_edges(e[:3])                                         # tail records before the graph
        ctx.tail_segment_end()                                                   # nothing open: a no-op
    with pytest.raises(_lib.HyperballError):
        gpu_ctx_factory(device=99)                                               # no such device


def test_golden_graphs(gpu_ctx_factory):
    gold = json.load(open(GOLD))
    for case in gold["graphs"]:
        ids, row_ptr, src = graphs.dense_from_tuples([tuple(e) for e in case["edges"]])
        for kw in (dict(), dict(chunk=4), dict(flags=_lib.HB_FLAG_NO_FRONTIER | _lib.HB_FLAG_NO_REORDER)):
            hc = HarmonicCentrality.calculate_dense(ids, row_ptr, src, **kw)
            assert hc.stats["passes"] == case["passes"], (case["name"], kw)
            assert _as_map(*hc.arrays()) == case["centrality_hex"], (case["name"], kw)


# ---- per-pass state parity --------------------------------------------------------------------
VARIANTS = {
    "default": dict(),
    "chunk4_multilevel": dict(chunk=4),
    "chunk16": dict(chunk=16, tune=(0, 1)),
    "no_frontier": dict(flags=_lib.HB_FLAG_NO_FRONTIER),
    "frontier_always": dict(tune=(0, 0, 101, 0, 0, 0, 1000000)),  # bitmap frontier whenever t > 0, never the sweep
    "frontier_always_multilevel": dict(chunk=8, tune=(0, 0, 101, 0, 0, 0, 1000000)),
    "sweep_small_direct_pass_stats": dict(chunk=32, flags=_lib.HB_FLAG_PASS_STATS, tune=(0, 0, 101, 5, 8, 8, 1)),
    "no_reorder_unroll4": dict(flags=_lib.HB_FLAG_NO_REORDER, tune=(0, 4)),
    "unfused": dict(flags=_lib.HB_FLAG_UNFUSED),
    "unfused_frontier_always": dict(flags=_lib.HB_FLAG_UNFUSED, tune=(0, 0, 101), chunk=8),
    "pass_stats": dict(flags=_lib.HB_FLAG_PASS_STATS),
    "few_blocks": dict(tune=(1,)),
    "host_plan": dict(flags=_lib.HB_FLAG_HOST_PLAN),
    "host_plan_chunk8_sweep": dict(flags=_lib.HB_FLAG_HOST_PLAN, chunk=8, tune=(0, 0, 101, 0, 0, 0, 1)),
    "lds_tile_experiment": dict(chunk=16, tune=(0, 0, 0, 6, 4, 0, 0, 256)),
    "sparse_always_multilevel": dict(chunk=8, tune=(0, 0, 101, 0, 0, 0, 1)),
    "sparse_always_banded": dict(chunk=16, tune=(0, 0, 101, 6, 4, 0, 1)),
    "no_sparse": dict(flags=_lib.HB_FLAG_NO_SPARSE),
    "no_init_pass": dict(flags=_lib.HB_FLAG_NO_INIT_PASS),  # pass 0 as a generic dense pass (the default streams the initial registers)
    "no_xcd_map": dict(flags=_lib.HB_FLAG_NO_XCD_MAP, chunk=16, tune=(0, 0, 0, 6, 4)),
    "xcd_slices_small_bands": dict(chunk=8, tune=(0, 0, 0, 4, 2)),
    "banded_chunks": dict(chunk=16, tune=(0, 0, 0, 6, 4)),
    "banded_frontier_small_direct": dict(chunk=32, tune=(0, 0, 101, 5, 8, 8)),
    # measurement switches of round 3: the older kernels stay bit-identical
    "old_per_tile_epilogue": dict(tune=(0, 0x100)),                                  # dense node rows: estimator/Kahan per quad, not once per row
    "frontier_always_chunk128": dict(chunk=128, tune=(0, 0, 101, 0, 0, 0, 1000000)), # rows with > 64 sources: two batches per hub chunk
    "frontier_always_pass_stats": dict(flags=_lib.HB_FLAG_PASS_STATS, tune=(0, 0, 101, 0, 0, 0, 1000000)),
    "sweep_general_seed_path": dict(chunk=8, tune=(0, 0x800, 101, 0, 0, 0, 1)),        # collect + expand + heavy also in the tail
    "frontier_always_slot_by_slot": dict(chunk=8, tune=(0, 0x2000, 101, 0, 0, 0, 1000000)),  # bitmap passes without packing the surviving sources
    "frontier_always_slot_by_slot_chunk128": dict(chunk=128, flags=_lib.HB_FLAG_PASS_STATS, tune=(0, 0x2000, 101, 0, 0, 0, 1000000)),
}


# round 5: the results travel in stages while the passes run (hb_api_pass.inc results_stage); bit 15 = a snapshot after EVERY pass
# whatever the graph's size, bit 17 = only the first snapshot, bit 16 = a final list of 16 entries (with bit 17: it overflows and the
# whole image is shipped instead)
VARIANTS["staged_results_every_pass"] = dict(tune=(0, 0x8000))
VARIANTS["staged_results_one_snapshot"] = dict(tune=(0, 0x28000))
VARIANTS["staged_results_sweep_tiny_list"] = dict(chunk=8, tune=(0, 0x38000, 101, 0, 0, 0, 1))
VARIANTS["staged_results_off"] = dict(tune=(0, 0x4000))
EXPECT_MODES = {"frontier_always": {0, 1}, "frontier_always_slot_by_slot": {0, 1}, "frontier_always_multilevel": {0, 1}, "sparse_always_multilevel": {0, 2},
                "long_tail_default": {0, 2}}
VARIANTS["long_tail_default"] = dict()
VARIANTS["long_tail_chunk8"] = dict(chunk=8)
VARIANTS["long_tail_staged_results"] = dict(tune=(0, 0x8000))
# round 5: the far tail as one workgroup (hb_tail.hip.h; off by default; hb_step lets it run ONE pass per launch, so every pass is
# compared): bit 21 = on, after a sweep pass that changed <= 4096 nodes with short reader lists; bit 22 = on, after any pass (right
# behind the dense passes: stale virtual bits, large dirty sets, lists that overflow and make it decline)
VARIANTS["long_tail_tail_kernel_after_any_pass"] = dict(tune=(0, 0x400000))
VARIANTS["long_tail_tail_kernel_after_any_pass_chunk4"] = dict(chunk=4, tune=(0, 0x400000))
VARIANTS["long_tail_tail_kernel_on"] = dict(tune=(0, 0x200000))
VARIANTS["tail_kernel_after_any_pass_sparse_always"] = dict(chunk=8, tune=(0, 0x400000, 101, 0, 0, 0, 1))


# round 6: hb_begin leaves the initial state (counters, Kahan words, sizes) to a LEAN pass 0 (hb_kernels.hip.h PassParams::rd_init) when
# that pass is the fused INIT launch of one rank; looking at the state right after hb_begin materialises it and pass 0 runs as before.
# "<variant>_lean": the same knobs WITHOUT that look, so the lean pass 0 itself runs and everything is compared after it (registers of
# all rows incl. the ones pass 0 left unchanged, every Kahan word and size, and - through pass 1 in its mode - the other buffer).
for _base in ("default", "chunk4_multilevel", "sparse_always_multilevel", "frontier_always", "old_per_tile_epilogue", "no_reorder_unroll4",
              "long_tail_default", "staged_results_every_pass", "no_xcd_map"):
    VARIANTS[_base + "_lean"] = VARIANTS[_base]
VARIANTS["full_init_switch"] = dict(tune=(0, 0x800000))  # hb_begin always writes the whole initial state (the pre-round-6 form)
# round 6: the transposed work-row graph (the sweep passes' reader lists) is built by ONE stable radix sort (hb_plan.hip gpu_transpose_rows);
# bit 25 = the atomic-scatter form it replaced, which remains the out-of-memory fallback: same passes, same bits either way
VARIANTS["sweep_transpose_by_scatter"] = dict(chunk=8, tune=(0, 0x2000000, 101, 0, 0, 0, 1))
VARIANTS["long_tail_transpose_by_scatter"] = dict(tune=(0, 0x2000000))
# round 6: a touched row of a sweep pass takes all its indices, then all their changed-bit words, then the needed gathers (hb_sweep.hip.h);
# bit 26 = the round-by-round loop it replaced (A/B form)
VARIANTS["pass0_level1_generic_kernel"] = dict(tune=(0, 0x8000000))  # bit 27: pass 0's first hub level through pass_kernel<.., INIT> (the form before init_level1_kernel)
VARIANTS["pass0_level1_chunk128"] = dict(chunk=128)                   # init_level1_kernel with rows of more than 64 sources: two batches
VARIANTS["sweep_rows_round_by_round"] = dict(chunk=8, tune=(0, 0x4000000, 101, 0, 0, 0, 1))
VARIANTS["long_tail_rows_round_by_round"] = dict(tune=(0, 0x4000000))
VARIANTS["sparse_always_chunk128"] = dict(chunk=128, tune=(0, 0, 101, 0, 0, 0, 1))  # sweep rows with more than 64 sources: two batches per hub chunk
VARIANTS["sparse_always_direct32"] = dict(chunk=32, tune=(0, 0, 101, 5, 8, 8, 1))    # node rows with more than 16 direct sources: several batches


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_per_pass_state_matches_oracle(gpu_ctx_factory, variant):
    # long_tail_*: R-MAT core + levelled-DAG tail (tens of passes: dense -> push masks -> worklists)
    g = synth.RmatGraph(13, 60_000, tail=(300, 800, 10)) if variant.startswith("long_tail") else synth.RmatGraph(13, 60_000)
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    kw = VARIANTS[variant]
    with gpu_ctx_factory(**kw) as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        ctx.begin()
        if not variant.endswith("_lean"):
            assert np.array_equal(ctx.registers(), o.registers())
            assert np.array_equal(ctx.sizes(), o.sizes())
        has = True
        t = 0
        modes = set()
        while has:
            has = ctx.step()
            ohas, ost = o.step(hbo.FRONTIER)
            assert has == ohas, t
            assert np.array_equal(ctx.registers(), o.registers()), "registers differ after pass %d" % t
            s, e = ctx.kahan()
            os_, oe = o.kahan()
            assert np.array_equal(s.view(np.uint64), os_.view(np.uint64)), "Kahan sum differs after pass %d" % t
            assert np.array_equal(e.view(np.uint64), oe.view(np.uint64)), "Kahan err differs after pass %d" % t
            assert np.array_equal(ctx.sizes(), o.sizes()), t
            assert ctx.state_hash() == o.state_hash(), t  # the checksum bench.py uses where n*64 bytes are too many to ship
            ps = ctx.pass_stats()[t]
            assert ps["changed"] == ost["changed"], t
            if ps["mode"] != 0:  # node rows with >= 1 gathered source (a split row counts only if a partial changed)
                assert ps["touched"] <= ost["touched"], t
            if kw.get("flags", 0) & _lib.HB_FLAG_PASS_STATS:
                if ps["mode"] == 1:  # counted edge by edge in the frontier kernel
                    assert ps["active_edges"] == ost["active_edges"], t
            else:  # A_t from the out-degree sum of the nodes that changed in pass t-1
                assert ps["active_edges"] == ost["active_edges"], t
            modes.add(ps["mode"])
            t += 1
        ctx.finish()
        vals, keep, k = o.finish()
        st = ctx.stats()
        _check_final(ctx, g.ids, t, vals, keep, st)
        if "staged_results" in variant:
            on = variant != "staged_results_off"
            assert (st["result_stages"] >= 1) == on, st
            if "every_pass" in variant or variant == "long_tail_staged_results":
                assert st["result_stages"] == t - 1 and st["result_list"] <= g.n, st  # (no snapshot after the pass that ends the loop)
            if "one_snapshot" in variant:
                assert st["result_stages"] == 1 and 16 < st["result_list"] <= g.n, st  # everything that moved after pass 0
            if "tiny_list" in variant:
                assert st["result_stages"] == 1 and st["result_list"] > 16, st         # ... and did not fit: the image went whole
            ranks = ctx.ranks()
            assert np.array_equal(ranks, hbo.rank_results(vals[keep])), variant          # the device image the ranks are cut from is final too
        if variant in EXPECT_MODES:
            assert EXPECT_MODES[variant] <= modes, (variant, modes)


def test_run_pipelines_the_convergence_tail_and_books_it_like_single_steps(gpu_ctx_factory):
    """hb_run keeps one pass queued ahead in the convergence tail [r5] (a pass guarded on the device by its predecessor's changed count
    goes into the stream before the host has read that count, so no host round trip separates the tail's passes): the loop must
    still end on the first pass that changes nothing (harmonic.rs:237-240), the guarded pass behind it must leave the state alone,
    and what hb_run books per pass must equal what one-pass-at-a-time stepping books - on the long-tail graph (tens of tail
    passes), with the pipeline on (default) and off (tune[1] bit 20), through sweep passes with the small and the general seed path."""
    g = synth.RmatGraph(13, 60_000, tail=(300, 800, 10))
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    T = o.run()
    vals, keep, k = o.finish()
    keys = ("pass", "changed", "active_edges", "touched", "mode")
    seen = {}
    for name, kw in (("pipelined", dict()), ("stepwise", dict(tune=(0, 0x100000))), ("pipelined_chunk8", dict(chunk=8)),
                     ("pipelined_staged_every_pass", dict(tune=(0, 0x8000)))):
        with gpu_ctx_factory(**kw) as ctx:
            ctx.load_dense(g.ids, g.row_ptr, g.src)
            for again in range(2):  # (a second run on the same context: the pipeline's buffers and events are reused)
                st = ctx.run()
                _check_final(ctx, g.ids, T, vals, keep, st)
                assert ctx.state_hash() == o.state_hash(), name            # registers + Kahan words as the LAST REAL pass left them
                assert (st["pipelined_passes"] > 10) == (name != "stepwise"), (name, st["pipelined_passes"])
                seen[name] = [tuple(ps[f] for f in keys) for ps in ctx.pass_stats()]
                assert len(seen[name]) == T and all(ps["ms_gpu"] > 0 for ps in ctx.pass_stats()), name
    assert seen["pipelined"] == seen["stepwise"] == seen["pipelined_staged_every_pass"]
    # ... and the same bookkeeping as stepping through the C ABI one pass at a time
    with gpu_ctx_factory() as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        ctx.begin()
        while ctx.step():
            pass
        ctx.finish()
        assert [tuple(ps[f] for f in keys) for ps in ctx.pass_stats()] == seen["pipelined"]


def test_tail_kernel_runs_whole_passes_from_work_lists(gpu_ctx_factory):
    """The far tail as ONE workgroup [r5] (hb_tail.hip.h): once a sweep pass has changed <= 4096 nodes with short reader lists, a single
    launch runs whole passes from work lists - seeds -> touched rows by level -> node rows -> carry-over / `+= 0.0` - keeps every
    bitmap exact by list, and goes on by itself until a pass changes nothing, up to 64 passes per launch (hb_run) or one (hb_step:
    the per-pass variants of test_per_pass_state_matches_oracle check registers / Kahan words / sizes after each of them).  Here:
    the same run through hb_run with the kernel on, off, and entered right after the dense passes (bit 22: stale virtual bits, a
    large Kahan-dirty set), on single- and multi-level chunk trees: same pass count, same per-pass changed counts / touched rows /
    A_t, same final list, state checksum = the oracle's; and the multi-kernel path takes over and hands back in mid-run (a pass
    budget of one launch is 64 passes; a second run on the same context re-collects its lists)."""
    g = synth.RmatGraph(13, 60_000, tail=(300, 800, 10))
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    T = o.run()
    vals, keep, k = o.finish()
    keys = ("pass", "changed", "active_edges", "touched")
    seen = {}
    # (the kernel is OFF by default - measured no faster than the launches it replaces; tune[1] bit 21 = on, bit 22 = on,
    # after any pass)
    for name, kw in (("on", dict(tune=(0, 0x200000))), ("off", dict()), ("after_any_pass", dict(tune=(0, 0x400000))), ("on_chunk8", dict(chunk=8, tune=(0, 0x200000))),
                     ("after_any_pass_chunk4_staged", dict(chunk=4, tune=(0, 0x408000)))):
        with gpu_ctx_factory(**kw) as ctx:
            ctx.load_dense(g.ids, g.row_ptr, g.src)
            for again in range(2):
                st = ctx.run()
                _check_final(ctx, g.ids, T, vals, keep, st)
                assert ctx.state_hash() == o.state_hash(), name
                ps = ctx.pass_stats()
                seen[name] = [tuple(p[f] for f in keys) for p in ps]
                in_kernel = sum(p["mode"] == 4 for p in ps)
                assert in_kernel == st["tail_kernel_passes"] and (in_kernel >= 10) == (name != "off"), (name, in_kernel)
                if "after_any_pass" in name:
                    assert in_kernel > seen_on, (in_kernel, seen_on)  # entered earlier than the default policy does
                elif name == "on":
                    seen_on = in_kernel
    assert seen["on"] == seen["off"]  # (touched rows included: the sweep kernels and the list kernel count the same rows)
    strip = lambda rows: [r[:3] for r in rows]  # (a bitmap pass counts touched rows its own way: compare changed counts and A_t there)
    assert strip(seen["on"]) == strip(seen["after_any_pass"]) == strip(seen["on_chunk8"]) == strip(seen["after_any_pass_chunk4_staged"])


def test_max_passes_is_reported_by_every_driver_of_the_tail(gpu_ctx_factory):
    """hb_options.max_passes: a run that needs more passes fails with HB_ERR_LIMIT - one pass at a time, with the tail pipeline (which
    queues a pass ahead and must not queue one beyond the limit) and with the single-workgroup kernel (whose launch budget is cut to
    what is left); with exactly enough passes it succeeds."""
    g = synth.RmatGraph(13, 60_000, tail=(300, 800, 10))
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    T = o.run()
    vals, keep, k = o.finish()
    for tune in ((), (0, 0x100000), (0, 0x200000), (0, 0x400000)):
        for limit in (T - 1, T - 2, T - 7):
            with gpu_ctx_factory(max_passes=limit, tune=tune) as ctx:
                ctx.load_dense(g.ids, g.row_ptr, g.src)
                with pytest.raises(_lib.HyperballError) as e:
                    ctx.run()
                assert e.value.code == _lib.HB_ERR_LIMIT, (tune, limit, str(e.value))
        with gpu_ctx_factory(max_passes=T, tune=tune) as ctx:
            ctx.load_dense(g.ids, g.row_ptr, g.src)
            _check_final(ctx, g.ids, T, vals, keep, ctx.run())


def test_tail_pipeline_when_a_late_change_would_ask_for_a_dense_pass(gpu_ctx_factory):
    """Found by tools/diff_fuzz.py on the MI355X (round 5): deep in the convergence tail ONE node with a large share of all out-links
    changes (a chain trickling into a hub whose counter is already large, so that it moves only now and then) - A_{t+1} jumps above
    the bitmap / dense thresholds, and the one-pass-at-a-time driver answers with a dense pass.  A pass of hb_run's pipeline is
    queued before that is known and must be a (device-guarded) sweep pass whatever the thresholds say; same values, same pass
    count, same per-pass changed counts."""
    chain = [(1000 + i, 1001 + i, 0) for i in range(60)]
    hub = 5000
    leaves = [(hub, 10_000 + j, 0) for j in range(3000)]
    background = [(20_000 + (j * 7) % 500, 20_000 + (j * 13 + 1) % 500, 0) for j in range(1500)]
    feed = [(20_000 + j, hub, 0) for j in range(400)]
    ids, row_ptr, src = graphs.dense_from_tuples(chain + [(1060, hub, 0)] + leaves + background + feed)
    o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
    T = o.run()
    vals, keep, k = o.finish()
    seen = {}
    for name, kw in (("pipelined", dict()), ("stepwise", dict(tune=(0, 0x100000))), ("pipelined_chunk8", dict(chunk=8)),
                     ("tail_kernel", dict(tune=(0, 0x200000))), ("tail_kernel_chunk8", dict(chunk=8, tune=(0, 0x200000)))):
        with gpu_ctx_factory(**kw) as ctx:
            ctx.load_dense(ids, row_ptr, src)
            st = ctx.run()
            _check_final(ctx, ids, T, vals, keep, st)
            assert ctx.state_hash() == o.state_hash(), name
            ps = ctx.pass_stats()
            seen[name] = [(p["pass"], p["changed"], p["active_edges"]) for p in ps]
            first_sweep = min(p["pass"] for p in ps if p["mode"] in (2, 4))
            late = [p["mode"] for p in ps if p["pass"] > first_sweep]
            if name == "stepwise":
                assert st["pipelined_passes"] == 0 and any(m not in (2, 4) for m in late), late  # the hub moved: a dense pass in the middle of the tail
            elif name.startswith("pipelined"):
                assert st["pipelined_passes"] >= 5 and all(m == 2 for m in late), (st["pipelined_passes"], late)
            else:  # the single-workgroup kernel runs these passes (the hub's 3000 readers fit its lists) - or hands one to the other path
                assert st["tail_kernel_passes"] >= 5 and all(m in (2, 4) for m in late), (st["tail_kernel_passes"], late)
    assert seen["pipelined"] == seen["stepwise"] == seen["pipelined_chunk8"] == seen["tail_kernel"] == seen["tail_kernel_chunk8"]


def test_salted_edge_records_match_faithful_oracle(gpu_ctx_factory):
    g = synth.RmatGraph(12, 30_000)
    e = g.edges(salt=1, salt_seed=3)
    fids, fvals, fst = hbo.faithful_run(e)
    graph = EdgeListGraph(e)
    hc = HarmonicCentrality.calculate(graph)
    ids, vals = hc.arrays()
    assert hc.stats["n"] == fst["n"] and hc.stats["m_unique"] == fst["m_unique"] and hc.stats["m_eff"] == fst["m_eff"]
    assert hc.stats["passes"] == fst["passes"]
    assert np.array_equal(ids, fids)
    assert np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
    # chunked hand-over, node set derived by the library
    with gpu_ctx_factory() as ctx:
        for part in np.array_split(e, 7):
            ctx.append_edges(part)
        ctx.finalize()
        ctx.run()
        ids2, vals2 = ctx.results()
    assert np.array_equal(ids2, fids) and np.array_equal(vals2.view(np.uint64), fvals.view(np.uint64))


def test_streamed_ingest_chunks_refusal_and_spill(gpu_ctx_factory):
    """hb_append_edges at its edges, reached with small inputs through hb_debug_set_ingest_limits: many small record chunks
    (pair keys are mapped chunk by chunk), a record-count refusal (the stream moves to the host path: the records come back
    from the device through the endpoint table) and an out-of-memory spill in the MIDDLE of a stream.  Input: the
    synthetic stream that the reference semantics reduce to exactly the clean graph (flagged-first pairs stay lost,
    later flagged duplicates are ignored) - so the oracle's dense run over the clean graph is the expected result, and
    the faithful oracle on the records says the same."""
    g = synth.RmatGraph(12, 30_000)
    total = g.stream_len(2)
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    T = o.run()
    ovals, keep, k = o.finish()
    recs = np.concatenate([sl.copy() for sl in g.stream(2, slab=5000)])
    fids, fvals, fst = hbo.faithful_run(recs)
    assert fst["n"] == g.n and fst["m_eff"] == g.m and np.array_equal(fids, g.ids[keep])
    assert np.array_equal(fvals.view(np.uint64), ovals[keep].view(np.uint64))
    cases = {"default": dict(), "chunks_of_4096": dict(chunk_records=4096), "chunks_of_1000_odd_batches": dict(chunk_records=1000),
             "refused_at_10000_records": dict(max_records=10_000), "oom_after_3_chunks": dict(chunk_records=4096, max_device_bytes=32768 * 20 + 3 * 4096 * 9),  # endpoint table + 3 chunks
             "oom_at_once": dict(max_device_bytes=1)}
    for name, lim in cases.items():
        with gpu_ctx_factory() as ctx:
            ctx.set_ingest_limits(**lim)
            slab = 3333 if "odd" in name else 5000
            for part in g.stream(2, slab=slab):
                ctx.append_edges(part)
            ctx.finalize()
            st = ctx.run()
            ids, vals = ctx.results()
            gi, grp, gsrc = ctx.graph()
        assert (st["n"], st["m_input"], st["m_eff"], st["m_unique"]) == (g.n, total, g.m, g.m + g.stream_lost_pairs(2)), name
        assert np.array_equal(gi, g.ids) and np.array_equal(grp, g.row_ptr) and np.array_equal(gsrc, g.src), name
        assert st["passes"] == T and np.array_equal(ids, g.ids[keep]) and np.array_equal(vals.view(np.uint64), ovals[keep].view(np.uint64)), name
        spilled = name.startswith(("refused", "oom"))
        assert (st["ingest_peak_bytes"] == 0) == spilled, (name, st["ingest_peak_bytes"])
        if not spilled:   # 9 B per record held (chunk granularity) + the endpoint table; 16 B per record while the pairs are sorted
            assert 9 * total <= st["ingest_peak_bytes"] < 25 * total + (64 << 20), (name, st["ingest_peak_bytes"])
    # hb_load_edges with more records than the device reduction takes: host ingest, same graph
    with gpu_ctx_factory() as ctx:
        ctx.set_ingest_limits(max_records=1000)
        ctx.load_edges(recs)
        st = ctx.stats()
        assert (st["n"], st["m_eff"], st["ingest_peak_bytes"]) == (g.n, g.m, 0)


def test_load_webgraph_from_edge_store(gpu_ctx_factory, tmp_path):
    """hb_load_webgraph: the native column reader (include/hb_webgraph.h) streams an on-disk edge store - three
    segments written by tests/tantivy_fixture.py - into the library; result = the faithful oracle on the same records."""
    from stract_amd import webgraph
    from tests import tantivy_fixture as tf
    g = synth.RmatGraph(12, 30_000)
    e = g.edges(salt=1, salt_seed=5)
    fids, fvals, fst = hbo.faithful_run(e)
    tf.write_edge_store(str(tmp_path / "edges"), [e[:7000], e[7000:7001], e[7001:]])
    for flags in (0, _lib.HB_FLAG_HOST_INGEST, _lib.HB_FLAG_HOST_PLAN):
        with gpu_ctx_factory(flags=flags) as ctx:
            webgraph.load_webgraph(ctx, str(tmp_path / "edges"), verify_crc=True)
            st = ctx.run()
            ids, vals = ctx.results()
        assert st["n"] == fst["n"] and st["m_unique"] == fst["m_unique"] and st["m_eff"] == fst["m_eff"] and st["passes"] == fst["passes"]
        assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
    # the many-slab hand-over (two alternating pinned buffers, reader thread ahead of the thread that feeds the library) on the same
    # small store: 1000-record slabs instead of 4 Mi (BASELINE-size stores take this path under bench.py's end-to-end leg)
    os.environ["HB_WEBGRAPH_SLAB_RECORDS"] = "1000"
    try:
        with gpu_ctx_factory() as ctx:
            webgraph.load_webgraph(ctx, str(tmp_path / "edges"), verify_crc=True)
            st = ctx.run()
            ids, vals = ctx.results()
        assert st["n"] == fst["n"] and st["m_input"] == len(e) and st["m_eff"] == fst["m_eff"] and st["passes"] == fst["passes"]
        assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
    finally:
        del os.environ["HB_WEBGRAPH_SLAB_RECORDS"]
    with gpu_ctx_factory() as ctx:
        with pytest.raises(_lib.HyperballError):
            webgraph.load_webgraph(ctx, str(tmp_path / "nothing_here"))
    # a damaged column file: the CRC check runs beside the streaming and must stop the load before hb_finalize; what was
    # appended is discarded (hb_discard_appended), so the SAME context then loads the intact store and gives the right result
    import shutil
    bad = tmp_path / "damaged" / "edges"
    shutil.copytree(str(tmp_path / "edges"), str(bad))
    victim = sorted(f for f in os.listdir(bad) if f.endswith(".col"))[-1]
    with open(bad / victim, "r+b") as f:
        f.seek(200)
        b = f.read(1)
        f.seek(200)
        f.write(bytes([b[0] ^ 0x40]))
    with gpu_ctx_factory() as ctx:
        with pytest.raises(_lib.HyperballError) as err:
            webgraph.load_webgraph(ctx, str(bad), verify_crc=True)
        assert "CRC mismatch" in str(err.value) and victim[:8] in str(err.value)
        webgraph.load_webgraph(ctx, str(tmp_path / "edges"), verify_crc=True)
        st = ctx.run()
        ids, vals = ctx.results()
        assert st["n"] == fst["n"] and st["m_input"] == len(e) and st["m_eff"] == fst["m_eff"]
        assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
        webgraph.load_webgraph(ctx, str(bad))       # without the check the damaged store loads (one flipped id bit: another graph)


def test_store_harmonic_writes_readable_stores(gpu_ctx_factory, tmp_path):
    """centrality/mod.rs:72-114 through the operator mirror: calculate -> store_harmonic -> both speedy_kv databases read
    back by the independent reader (tests/speedy_kv_reader.py) hold the library's results and the reference's rank order"""
    from stract_amd.harmonic import store_harmonic
    from tests import speedy_kv_reader as kv
    g = synth.RmatGraph(12, 30_000)
    hc = HarmonicCentrality.calculate_dense(g.ids, g.row_ptr, g.src)
    ids, vals = hc.arrays()
    assert np.array_equal(hc.ranks(), hbo.rank_results(vals))
    store_harmonic(hc, str(tmp_path))
    cen = kv.Db(str(tmp_path / "harmonic"), "f64", str(tmp_path))
    rnk = kv.Db(str(tmp_path / "harmonic_rank"), "u64", str(tmp_path))
    assert len(cen) == len(rnk) == hc.len()
    ints = kv.ids_to_ints(ids)
    got = dict(cen.items())
    assert all(np.float64(got[i]).view(np.uint64) == np.float64(v).view(np.uint64) for i, v in zip(ints, vals.tolist()))
    assert dict(rnk.items()) == dict(zip(ints, hc.ranks().tolist()))
    # top_nodes (centrality/mod.rs:33-52) from the stored ranks = the library's top()
    by_rank = sorted(rnk.items(), key=lambda kv_: kv_[1])[:10]
    assert [i for i, _ in by_rank] == [ints[j] for j in np.argsort(hc.ranks(), kind="stable")[:10].tolist()]
    fixture = HarmonicCentrality.calculate(graphs.fixture_graph())
    store_harmonic(fixture, str(tmp_path / "fixture"))
    r = kv.Db(str(tmp_path / "fixture" / "harmonic_rank"), "u64", str(tmp_path))
    assert (r.get(graphs.C), r.get(graphs.A), r.get(graphs.B), r.get(graphs.D)) == (0, 1, 2, None)  # harmonic.rs:465-473


def test_store_from_the_context_equals_the_host_sorted_store(gpu_ctx_factory, tmp_path):
    """hb_store_harmonic_results [r5]: the key order of both databases from a 136-bit radix sort on the device (bincode key BYTES: a
    class byte, then the integer little endian - not NodeID order) must give the same files, byte for byte, as hb_store_harmonic's
    host sort of the same arrays.  Ids of every bincode length class (1, 3, 5, 9, 17 bytes, both sides of every boundary) and an
    R-MAT graph with hashed ids."""
    import glob
    special = [1, 5, 250, 251, 252, 65535, 65536, 65537, (1 << 32) - 1, 1 << 32, (1 << 32) + 9, (1 << 64) - 1, 1 << 64, (1 << 64) + 1,
               (1 << 100) + 7, (1 << 127) + 3, (1 << 128) - 1, 254, 253, 1 << 16, 3 << 40]
    chain = graphs.dense_from_tuples([(a, b, 0) for a, b in zip(special, special[1:])] + [(special[-1], special[0], 0), (special[3], special[9], 0)])
    rmat = synth.RmatGraph(12, 30_000)
    for case, (ids, row_ptr, src) in enumerate((chain, (rmat.ids, rmat.row_ptr, rmat.src))):
        with gpu_ctx_factory() as ctx:
            ctx.load_dense(ids, row_ptr, src)
            ctx.run()
            rid, rvals = ctx.results()
            ranks = ctx.ranks()
            a, b = tmp_path / ("dev%d" % case), tmp_path / ("host%d" % case)
            ctx.store_harmonic(str(a))
            _lib.store_harmonic(str(b), rid, rvals, ranks)
            assert len(rid) > (10 if case == 0 else 1000)
            for db in ("harmonic", "harmonic_rank"):
                for ext in ("blobs", "bid", "ids", "blm"):
                    fa, fb = glob.glob(str(a / db / ("*." + ext))), glob.glob(str(b / db / ("*." + ext)))
                    assert len(fa) == 1 and len(fb) == 1, (db, ext)
                    assert open(fa[0], "rb").read() == open(fb[0], "rb").read(), (case, db, ext)
            with pytest.raises(_lib.HyperballError):
                ctx.store_harmonic(str(a))  # the directory holds databases now: refused like the host entry point


@pytest.mark.parametrize("image", ["one_entry_per_node", "compact"])
def test_result_ranks_match_store_harmonic_order(gpu_ctx_factory, image):
    # centrality/mod.rs:92-103: harmonic_rank = position by (Reverse(total_cmp(centrality)), NodeID)
    # image = compact [r6]: results shipped in stages (forced here; the default from 2^20 nodes on) keep one image entry per node WITH
    # in-edges only; list, ranks and top-k must come out of it exactly as out of the one-entry-per-node image
    g = synth.RmatGraph(13, 60_000)
    kw = dict(tune=(0, 0x8000)) if image == "compact" else {}
    _factory = gpu_ctx_factory
    gpu_ctx_factory = lambda: _factory(**kw)  # noqa: E731
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    while o.step(hbo.FRONTIER)[0]:
        pass
    ovals, okeep, ok = o.finish()
    indeg = np.diff(g.row_ptr)
    assert (indeg == 0).sum() > 100 and ok < g.n  # the graph has nodes the compact image leaves out
    with gpu_ctx_factory() as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        ctx.run()
        ids, vals = ctx.results()
        ranks = ctx.ranks()
        assert (ctx.stats()["result_stages"] >= 1) == (image == "compact")
    assert np.array_equal(ids, g.ids[okeep]) and np.array_equal(vals.view(np.uint64), ovals[okeep].view(np.uint64))
    assert len(ranks) == len(vals) and sorted(ranks.tolist()) == list(range(len(vals)))
    assert np.array_equal(ranks, hbo.rank_results(vals))
    assert len(np.unique(vals)) < len(vals)  # ties exist: the NodeID tie-break is exercised
    # top_nodes(TopNodes::Top(k)) (centrality/mod.rs:33-52) = the first k of that order
    by_rank = np.argsort(ranks, kind="stable")
    with gpu_ctx_factory() as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        ctx.run()
        for k in (0, 1, 1000, len(vals), len(vals) + 5):
            tids, tvals = ctx.top(k)
            kk = min(k, len(vals))
            assert len(tvals) == kk and np.array_equal(tids, ids[by_rank[:kk]]) and np.array_equal(tvals, vals[by_rank[:kk]])
        assert np.all(np.diff(ctx.top(1000)[1]) <= 0)
    hc = HarmonicCentrality.calculate(graphs.fixture_graph())
    with gpu_ctx_factory() as ctx:
        ctx.load_edges(graphs.fixture_graph().host_edges())
        ctx.run()
        assert ctx.ranks().tolist() == [1, 2, 0]  # C > A > B (harmonic.rs:465-473)
        assert hc.len() == 3
    # a run that ends before any snapshot was taken (2 nodes, one edge): the compact image then goes whole at hb_finish
    with gpu_ctx_factory() as ctx:
        e = np.zeros(1, dtype=_lib.EDGE)
        e["from"]["lo"], e["to"]["lo"] = 5, 9
        ctx.load_edges(e)
        ctx.run()
        i2, v2 = ctx.results()
        assert [int(x) for x in i2["lo"]] == [9] and v2.tolist() == [1.0] and ctx.ranks().tolist() == [0]
        t2 = ctx.top(5)
        assert [int(x) for x in t2[0]["lo"]] == [9] and t2[1].tolist() == [1.0]
    # ... and a graph without any edge: pass 0 changes nothing, no snapshot is ever taken, the image has no entry at all
    with gpu_ctx_factory() as ctx:
        ids3 = np.zeros(3, dtype=_lib.U128)
        ids3["lo"] = [2, 4, 6]
        ctx.load_dense(ids3, np.zeros(4, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
        ctx.run()
        assert len(ctx.results()[1]) == 0 and len(ctx.ranks()) == 0 and len(ctx.top(3)[1]) == 0 and ctx.stats()["result_stages"] == 0


def test_gpu_ingest_equals_host_ingest(gpu_ctx_factory):
    """hb_ingest.hip (rocPRIM sorts on the device) against hb_host.cpp and the plain-Python statement of
    store.rs:297-357 + harmonic.rs:131: node set, first-occurrence de-duplication, flag filter, CSR."""
    g = synth.RmatGraph(12, 30_000)
    e = g.edges(salt=1, salt_seed=11)
    cases = [(e, None), (e[:0], None), (e[:1], None)]
    # explicit node list (any order, duplicates, one endpoint of some records missing from it)
    ids_h, rp_h, src_h, mu_h = _lib.host_ingest(e)
    cases.append((e, np.concatenate([ids_h[::-1], ids_h[:7]])))
    cases.append((e, ids_h[: len(ids_h) // 2]))
    for edges, nodes in cases:
        got = []
        for flags in (0, _lib.HB_FLAG_HOST_INGEST):
            with gpu_ctx_factory(flags=flags) as ctx:
                ctx.load_edges(edges, nodes)
                st = ctx.stats()
                got.append((ctx.graph(), st["n"], st["m_unique"], st["m_eff"], st["m_input"]))
        (ga, *sa), (gb, *sb) = got
        assert sa == sb
        for x, y in zip(ga, gb):
            assert np.array_equal(x, y)
        ref = _lib.host_ingest(edges, nodes)
        assert np.array_equal(ga[0], ref[0]) and np.array_equal(ga[1], ref[1]) and np.array_equal(ga[2], ref[2])
        assert sa[1] == ref[3]


def test_hub_rows_and_isolated_nodes(gpu_ctx_factory):
    # a 3000-source star (multi-level virtual rows at chunk 8), a long chain (many passes in
    # frontier mode) and nodes that only appear on flagged edges (count in n, no output row)
    tuples = [(i, 1, 0) for i in range(2, 3002)] + [(5000 + i, 5001 + i, 0) for i in range(300)]
    tuples += [(1, 5000, 0), (9001, 9002, graphs.NOFOLLOW), (9003, 1, graphs.TAG)]
    e = EdgeListGraph.from_tuples(tuples)
    fids, fvals, fst = hbo.faithful_run(e.host_edges())
    for kw in (dict(chunk=8), dict(), dict(chunk=8, flags=_lib.HB_FLAG_UNFUSED), dict(chunk=8, tune=(0, 0, 101, 0, 0, 0, 1)),
               dict(flags=_lib.HB_FLAG_NO_SPARSE)):
        hc = HarmonicCentrality.calculate(e, **kw)
        if not kw.get("flags"):
            assert any(ps["mode"] == 2 for ps in hc.pass_stats)  # the worklist-driven tail ran
        ids, vals = hc.arrays()
        assert hc.stats["passes"] == fst["passes"] and hc.stats["n"] == fst["n"]
        assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64)), kw


def test_sweep_seeds_with_very_long_reader_lists(gpu_ctx_factory):
    """A hub SOURCE that keeps changing late: a chain with private feeders drips new elements into H for a dozen passes, H links to K
    leaves, everything else is quiet - so the data-driven (sweep) passes get a seed with K readers.  K = 3000: <= 4096 nodes changed,
    one launch collects and expands, H's list is walked by the whole wave (> 64 entries); K = 5000: the general path, H goes to the
    grid-wide expansion (> 4096 readers).  Found by measuring block coverage of the kernels on the interpreted device sources
    (tools/simt_coverage.py): no other test reached those branches."""
    for K in (3000, 5000):
        L = 14
        H, chain0, feed0, leaf0 = 1, 10, 1000, 100000
        tuples = [(chain0 + i, chain0 + i + 1, 0) for i in range(L - 1)] + [(chain0 + L - 1, H, 0)]
        tuples += [(feed0 + 8 * i + j, chain0 + i, 0) for i in range(L) for j in range(6)]
        tuples += [(H, leaf0 + k, 0) for k in range(K)]
        tuples += [(leaf0 + k, leaf0 + K + (k % 7), 0) for k in range(0, K, 3)]   # a few leaves are read further on
        e = EdgeListGraph.from_tuples(tuples)
        fids, fvals, fst = hbo.faithful_run(e.host_edges())
        for kw in (dict(tune=(0, 0, 101, 0, 0, 0, 1)), dict(tune=(0, 0x800, 101, 0, 0, 0, 1)), dict()):
            hc = HarmonicCentrality.calculate(e, **kw)
            ids, vals = hc.arrays()
            assert hc.stats["passes"] == fst["passes"] and hc.stats["n"] == fst["n"], (K, kw)
            assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64)), (K, kw)
            if kw:
                sweeps = [ps for ps in hc.pass_stats if ps["mode"] == 2]
                assert len(sweeps) >= 8 and max(ps["changed"] for ps in sweeps) >= K, (K, kw)  # H (and with it its K leaves) changed in a sweep pass


def test_randomized_graphs_and_knobs(gpu_ctx_factory):
    """Random small graphs of several shapes x random planner / mode knobs: final list, pass count and
    registers against the oracle."""
    rng = np.random.default_rng(977)
    flag_pool = [0, 0, _lib.HB_FLAG_NO_REORDER, _lib.HB_FLAG_NO_XCD_MAP, _lib.HB_FLAG_UNFUSED, _lib.HB_FLAG_NO_SPARSE,
                 _lib.HB_FLAG_NO_FRONTIER, _lib.HB_FLAG_PASS_STATS]
    for case in range(40):
        kind, edges = graphs.random_graph(rng)
        ids, row_ptr, src = graphs.dense_from_tuples(edges) if edges else (np.zeros(0, _lib.U128), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
        o, T, vals, keep, k = _oracle_dense(ids, row_ptr, src)
        chunk = int(rng.choice([4, 8, 16, 64]))
        tune = (int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2, 4])), int(rng.choice([0, 30, 101])), int(rng.integers(4, 9)),
                int(rng.integers(1, 9)), int(rng.integers(0, chunk + 1)), int(rng.choice([0, 1, 4])))
        flags = int(rng.choice(flag_pool)) | int(rng.choice(flag_pool))
        with gpu_ctx_factory(flags=flags, chunk=chunk, tune=tune) as ctx:
            ctx.load_dense(ids, row_ptr, src)
            st = ctx.run()
            what = (case, kind, len(ids), len(src), chunk, tune, flags)
            assert st["passes"] == T, what
            gids, gvals = ctx.results()
            assert np.array_equal(gids, ids[keep]), what
            assert np.array_equal(gvals.view(np.uint64), vals[keep].view(np.uint64)), what
            if len(ids):
                assert np.array_equal(ctx.registers(), o.registers()), what


# ---- AMPC operator ------------------------------------------------------------------------------------
def test_ampc_counter_table_upsert_semantics(gpu_ctx_factory):
    """include/hb_ampc.h: batch_set / batch_get / batch_upsert(HyperLogLog64Upsert) on a GPU-resident table shard
    against a dict that applies the pairs in order exactly like dht/store.rs:159-190 + upsert.rs:67-89."""
    from stract_amd import ampc
    rng = np.random.default_rng(23)
    model = {}
    keyspace = np.zeros(400, dtype=_lib.U128)
    keyspace["lo"] = rng.integers(0, 1 << 63, 400, dtype=np.uint64)
    keyspace["hi"] = rng.integers(0, 1 << 63, 400, dtype=np.uint64)
    kid = lambda k: (int(k["hi"]) << 64) | int(k["lo"])
    with ampc.CounterTable(capacity_hint=16) as tab:
        for step in range(12):
            n = int(rng.integers(1, 3000))
            idx = rng.integers(0, len(keyspace), n)  # many repeated keys inside one batch
            keys = keyspace[idx]
            vals = graphs.random_registers(rng, n)
            if step % 4 == 3:  # batch_set: later pairs win
                tab.batch_set(keys, vals)
                for k, v in zip(keys, vals):
                    model[kid(k)] = v.copy()
                continue
            acts = tab.batch_upsert(keys, vals)
            want = []
            for k, v in zip(keys, vals):
                old = model.get(kid(k))
                if old is None:
                    model[kid(k)] = v.copy()
                    want.append(ampc.INSERTED)
                else:
                    merged = np.maximum(old, v)  # HyperLogLog::merge, hyperloglog.rs:4531-4535
                    want.append(ampc.MERGED if not np.array_equal(merged, old) else ampc.NO_CHANGE)
                    model[kid(k)] = merged
            assert acts.tolist() == want, step
            assert len(tab) == len(model)
            got, found = tab.batch_get(keyspace)
            for k, g_, f in zip(keyspace, got, found):
                assert f == (kid(k) in model)
                assert np.array_equal(g_, model.get(kid(k), np.zeros(64, np.uint8)))
        # one pass of the distributed algorithm's counter update on a small graph == one dense pass of the oracle
        g = synth.RmatGraph(9, 3000)
        o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    with ampc.CounterTable() as prev, ampc.CounterTable() as nxt:
        regs0 = o.registers()
        prev.batch_set(g.ids, regs0)   # setup_counters, mapper.rs:64-88
        nxt.batch_set(g.ids, regs0)
        dst = np.repeat(np.arange(g.n), np.diff(g.row_ptr).astype(np.int64))
        old, _ = prev.batch_get(g.ids[g.src])                          # get_old_counters
        acts = nxt.batch_upsert(g.ids[dst], old)                       # update_counters
        o.step(0)
        got, _ = nxt.batch_get(g.ids)
        assert np.array_equal(got, o.registers())
        changed = np.zeros(g.n, dtype=bool)
        changed[dst[acts == ampc.MERGED]] = True
        assert np.array_equal(changed, np.any(o.registers() != regs0, axis=1))


def test_ampc_device_key_index_growth_duplicates_and_absent_keys(gpu_ctx_factory):
    """Round 5: the key -> slot index of the counter shard is a device hash table (hb_table.hip.h, the ingest's endpoint table).
    A table created for 4 keys takes 40 000 (the index is rebuilt several times, ids kept), batches in which one key occurs
    hundreds of times (its pairs must be applied in batch order: the action codes say so), keys that differ in one half only,
    consecutive small integers (no clustering), lookups of keys that were never stored, an empty batch, and - after all that -
    every stored counter equals the model's."""
    from stract_amd import ampc
    rng = np.random.default_rng(5)
    kid = lambda k: (int(k["hi"]) << 64) | int(k["lo"])
    space = np.zeros(40_000, dtype=_lib.U128)
    space["lo"][:20_000] = np.arange(20_000, dtype=np.uint64)            # consecutive integers
    space["lo"][20_000:30_000] = 7                                        # same low half, different high halves
    space["hi"][20_000:30_000] = np.arange(1, 10_001, dtype=np.uint64)
    space["lo"][30_000:] = rng.integers(0, 1 << 63, 10_000, dtype=np.uint64)
    space["hi"][30_000:] = rng.integers(0, 1 << 63, 10_000, dtype=np.uint64)
    model = {}
    with ampc.CounterTable(capacity_hint=4) as tab:
        assert len(tab) == 0
        tab.batch_set(space[:0], np.zeros((0, 64), np.uint8))             # an empty batch is fine
        got, found = tab.batch_get(space[:100])
        assert not found.any() and not got.any()                           # nothing stored yet: default counters
        for step, n in enumerate((1, 70, 5_000, 60_000, 9_000)):
            hot = rng.integers(0, len(space), 3)                           # three keys take a fifth of the batch
            idx = np.where(rng.random(n) < 0.2, hot[rng.integers(0, 3, n)], rng.integers(0, min(len(space), 40 * n + 10), n))
            keys, vals = space[idx], graphs.random_registers(rng, n)
            acts = tab.batch_upsert(keys, vals)
            want = np.empty(n, dtype=np.uint8)
            for j, (k, v) in enumerate(zip(keys, vals)):
                old = model.get(kid(k))
                if old is None:
                    model[kid(k)] = v.copy()
                    want[j] = ampc.INSERTED
                else:
                    merged = np.maximum(old, v)
                    want[j] = ampc.MERGED if not np.array_equal(merged, old) else ampc.NO_CHANGE
                    model[kid(k)] = merged
            assert np.array_equal(acts, want), step
            assert len(tab) == len(model), step
        got, found = tab.batch_get(space)
        stored = np.array([kid(k) in model for k in space])
        assert np.array_equal(found, stored)
        want = np.stack([model.get(kid(k), np.zeros(64, np.uint8)) for k in space])
        assert np.array_equal(got, want)
        with pytest.raises(_lib.HyperballError):
            tab.lib.hbu_batch_upsert.argtypes  # noqa: B018 (attribute exists)
            tab._check(tab.lib.hbu_batch_upsert(tab.h, None, None, 5, None))  # NULL with a count: refused, table untouched
        assert len(tab) == len(model)


# ---- device planner ------------------------------------------------------------------------------
def test_device_plan_equals_host_plan(gpu_ctx_factory):
    """hb_plan.hip (rocPRIM sorts / scans on the device) must reproduce build_plan() of hb_host.cpp entry for entry:
    device order, work-row offsets, every source list (hub chunk trees, slice cuts, XCD groups), level boundaries."""
    graphs_ = [synth.RmatGraph(13, 60_000), synth.RmatGraph(12, 30_000, tail=(300, 800, 10)), synth.RmatGraph(15, 400_000)]
    star = graphs.dense_from_tuples([(i, 1, 0) for i in range(2, 3002)] + [(5000 + i, 5001 + i, 0) for i in range(300)] + [(1, 5000, 0)])
    knobs = [dict(), dict(chunk=4), dict(chunk=8, tune=(0, 0, 0, 4, 2)), dict(chunk=16, tune=(0, 0, 0, 6, 4)),
             dict(flags=_lib.HB_FLAG_NO_XCD_MAP, chunk=16, tune=(0, 0, 0, 6, 4)), dict(flags=_lib.HB_FLAG_NO_REORDER),
             dict(chunk=32, tune=(0, 0, 0, 5, 8, 8)), dict(chunk=64, tune=(0, 0, 0, 1)), dict(chunk=256)]
    cases = [(g.ids, g.row_ptr, g.src) for g in graphs_] + [star]
    empty = (np.zeros(0, _lib.U128), np.zeros(1, np.uint64), np.zeros(0, np.uint32))
    for gi, (ids, row_ptr, src) in enumerate(cases + [empty]):
        for kw in knobs:
            with gpu_ctx_factory(**kw) as ctx:
                ctx.load_dense(ids, row_ptr, src)
                dev = ctx.plan()
                st = ctx.stats()
            host = _lib.host_plan(row_ptr, src, flags=kw.get("flags", 0), chunk=kw.get("chunk", 0), tune=kw.get("tune", ()))
            what = (gi, kw)
            n = len(ids)
            assert dev["n_pad"] == host["n_pad"] and dev["nv"] == host["nv"], what
            assert np.array_equal(dev["level_begin"], host["level_begin"]), what
            assert np.array_equal(dev["order"][:n], host["order"][:n]), what
            assert np.all(dev["order"][n:] == 0xFFFFFFFF), what
            assert np.array_equal(dev["row_ptr"], host["row_ptr"]), what
            assert np.array_equal(dev["src"], host["src"]), what
            assert st["level1_edges"] + st["direct_edges"] == len(src), what
    # destination partition: the device planner lays the rows out as `world` owner slices like the host planner does
    # (hb_host_plan with tune[7] = world); logical ranks without a communicator keep the id order inside a slice.
    # The rank is handed ALL in-edges and keeps its own rows' (gpu_keep_owned_rows), or only its own to begin with.
    g = graphs_[0]
    for world in (2, 3):
        for rank in range(world):
            rp_own, src_own = dist.partition_dense_by_dest(g.row_ptr, g.src, rank, world)
            host = _lib.host_plan(rp_own, src_own, flags=_lib.HB_FLAG_NO_REORDER, chunk=16, tune=(0, 0, 0, 7, 4, 0, 0, world))
            for rp_in, src_in in ((rp_own, src_own), (g.row_ptr, g.src)):
                with gpu_ctx_factory(rank=rank, world_size=world, flags=_lib.HB_FLAG_NO_RCCL | _lib.HB_FLAG_DEST_PARTITION, chunk=16,
                                     tune=(0, 0, 0, 7, 4)) as ctx:
                    ctx.load_dense(g.ids, rp_in, src_in)
                    dev = ctx.plan()
                    assert ctx.stats()["m_eff"] == len(src_own)
                what = (world, rank, len(src_in))
                assert dev["n_pad"] == host["n_pad"] and dev["nv"] == host["nv"], what
                assert np.array_equal(dev["level_begin"], host["level_begin"]), what
                assert np.array_equal(dev["order"], host["order"]), what
                assert np.array_equal(dev["row_ptr"], host["row_ptr"]) and np.array_equal(dev["src"], host["src"]), what


def test_c2_device_ingest_and_plan_equal_host(gpu_ctx_factory):
    """The two ingests and the two planners must stay entry-for-entry equal (hb_ingest.hip / hb_plan.hip vs hb_host.cpp) - also
    at BASELINE configs[1] size (1.1 M hosts / 20.7 M edges; 24.9 M raw records with flagged-first pairs and duplicates), where
    64-bit offsets, multi-chunk streams and every planner stage see real sizes."""
    cfg = synth.CONFIGS["C2"]
    g = synth.RmatGraph(cfg["scale"], cfg["m"])
    recs = np.zeros(g.stream_len(2), dtype=_lib.EDGE)
    assert g.stream_fill(recs, 0, 2) == len(recs)
    hi, hrp, hsrc, hmu = _lib.host_ingest(recs)
    assert np.array_equal(hi, g.ids) and np.array_equal(hrp, g.row_ptr) and np.array_equal(hsrc, g.src)
    with gpu_ctx_factory() as ctx:
        ctx.set_ingest_limits(chunk_records=1 << 22)          # six chunks
        for part in np.array_split(recs, 5):
            ctx.append_edges(part)
        ctx.finalize()
        st = ctx.stats()
        gi, grp, gsrc = ctx.graph()
        dev = ctx.plan()
    assert (st["n"], st["m_unique"], st["m_eff"]) == (g.n, hmu, g.m)
    # 9 B per record held + the endpoint table (sized for two new ids per record of a 4 Mi-record slab), 16 B per record in the sort
    assert 0 < st["ingest_peak_bytes"] < 25 * len(recs) + (1 << 30), st["ingest_peak_bytes"]
    assert np.array_equal(gi, hi) and np.array_equal(grp, hrp) and np.array_equal(gsrc, hsrc)
    host = _lib.host_plan(g.row_ptr, g.src)
    assert dev["n_pad"] == host["n_pad"] and dev["nv"] == host["nv"] and np.array_equal(dev["level_begin"], host["level_begin"])
    assert np.array_equal(dev["order"][:g.n], host["order"][:g.n])
    assert np.array_equal(dev["row_ptr"], host["row_ptr"]) and np.array_equal(dev["src"], host["src"])


# ---- edge-partition mode ------------------------------------------------------------------------
@pytest.mark.parametrize("world,mode", [(2, "edge"), (3, "edge"), (2, "edge_changed"), (4, "edge_changed"), (2, "dest"), (3, "dest"), (4, "dest"),
                                        (2, "dest_changed"), (4, "dest_changed")])
def test_logical_ranks_on_one_device(gpu_ctx_factory, world, mode):
    """SURVEY.md §8(e) caveat: R logical ranks on one device, the collective emulated by
    hb_debug_exchange (edge partition: all-reduce(max); destination partition: all-gather of the
    owned slices); every rank must reproduce the single-GPU result bit for bit."""
    g = synth.RmatGraph(12, 40_000)
    o, T, vals, keep, k = _oracle_dense(g.ids, g.row_ptr, g.src)
    flags = _lib.HB_FLAG_NO_RCCL | (_lib.HB_FLAG_DEST_PARTITION if mode.startswith("dest") else 0)
    flags |= _lib.HB_FLAG_CHANGED_ONLY if mode.endswith("_changed") else 0  # packed changed counters instead of whole slices / all rows
    split = dist.partition_dense_by_dest if mode.startswith("dest") else dist.partition_dense
    ctxs = []
    try:
        for r in range(world):
            c = gpu_ctx_factory(rank=r, world_size=world, flags=flags, chunk=16, tune=(0, 0, 0, 7, 4))
            rp, src = split(g.row_ptr, g.src, r, world)
            c.load_dense(g.ids, rp, src)
            c.begin()
            ctxs.append(c)
        has, t = True, 0
        while has:
            for c in ctxs:
                c.step_local()
            _lib.Context.exchange(ctxs, 0)
            flags_out = [c.step_finish() for c in ctxs]
            assert len(set(flags_out)) == 1
            has = flags_out[0]
            for c in ctxs[1:]:
                assert np.array_equal(c.registers(), ctxs[0].registers())
            t += 1
        assert t == T
        assert np.array_equal(ctxs[0].registers(), o.registers())
        _lib.Context.exchange(ctxs, 1)
        for c in ctxs:
            c.finish()
            _check_final(c, g.ids, T, vals, keep, c.stats())
        if mode.startswith("edge"):
            # bytes a rank receives over the run: the full all-reduce moves 2 (w-1)/w n 64 B in each of the T passes; the
            # changed-only form only the union of the locally changed rows (+ the ranks' bitmaps): fewer bytes in every pass
            n_pad = (g.n + 63) // 64 * 64
            full = T * 2 * (world - 1) * n_pad * 64 // world
            wb = ctxs[0].stats()["wire_bytes"]
            assert (wb == 0) if mode == "edge" else (0 < wb < 0.8 * full), (mode, wb, full)
    finally:
        for c in ctxs:
            c.close()


def test_packed_exchange_is_six_bits_per_register(gpu_ctx_factory):
    """[r6] The destination partition's changed-only exchange ships a counter as 48 bytes: 6 bits per register (hb_aux.hip.h
    pack6_quarter - exact, because a register of this path is 0, 1..58 or 65: hyperloglog.rs:4385-4396 shifts the hash left by six
    before counting leading zeros).  Two logical ranks, a graph that contains the one register value above 58 (NodeIDs whose low 64
    bits are 0 hash to 0: p = 65, code 63) spread by the merges: same registers after every pass and same final list as the oracle,
    and exactly 3/4 of the counter bytes of the 64-byte form (experiments build, tune[1] bit 24) on the wire."""
    g = synth.RmatGraph(11, 14_000)
    lcg = graphs.lcg_graph(n=300, m=2400, seed=7)
    zero_lo = [0, 1 << 64, 5 << 64]  # three ids with low 64 bits 0: their counters start with a register of 65
    ids, row_ptr, src = graphs.dense_from_tuples([(zero_lo[f % 3] if f % 50 == 0 else f + (f << 70), zero_lo[t % 3] if t % 50 == 0 else t + (t << 70)) for f, t in lcg])
    o, T, vals, keep, k = _oracle_dense(ids, row_ptr, src)
    assert int(o.registers().max()) == 65 and not np.isin(o.registers(), np.arange(59, 65)).any()
    world = 2
    n_pad_slices = None
    wires = {}
    for name, tune in (("six_bit", ()), ("bytes", (0, 0x1000000))):
        ctxs = []
        try:
            for r in range(world):
                c = gpu_ctx_factory(rank=r, world_size=world, flags=_lib.HB_FLAG_NO_RCCL | _lib.HB_FLAG_DEST_PARTITION | _lib.HB_FLAG_CHANGED_ONLY, tune=tune)
                rp, sr = dist.partition_dense_by_dest(row_ptr, src, r, world)
                c.load_dense(ids, rp, sr)
                c.begin()
                ctxs.append(c)
            has, t = True, 0
            while has:
                for c in ctxs:
                    c.step_local()
                _lib.Context.exchange(ctxs, 0)
                has = [c.step_finish() for c in ctxs][0]
                for c in ctxs:
                    assert np.array_equal(c.registers(), ctxs[0].registers())
                t += 1
            assert t == T and np.array_equal(ctxs[0].registers(), o.registers())
            _lib.Context.exchange(ctxs, 1)
            for c in ctxs:
                c.finish()
                _check_final(c, ids, T, vals, keep, c.stats())
            wires[name] = [int(c.stats()["wire_bytes"]) for c in ctxs]
            n_pad_slices = [int(c.stats()["work_rows"]) for c in ctxs]
        finally:
            for c in ctxs:
                c.close()
    # per rank: wire = T x (bitmap bytes of the foreign slices) + bytes per row x (foreign changed rows over the run)
    for r in range(world):
        rows64 = wires["bytes"][r]
        rows48 = wires["six_bit"][r]
        assert rows48 < rows64
        # the bitmap part is the same in both runs: (rows64 - B) * 3 == (rows48 - B) * 4  =>  B = 4 rows48 - 3 rows64
        b = 4 * rows48 - 3 * rows64
        assert b >= 0 and (rows64 - b) % 64 == 0 and (rows48 - b) % 48 == 0 and (rows64 - b) // 64 == (rows48 - b) // 48, (rows48, rows64, b)


def test_dest_partition_ignores_foreign_records(gpu_ctx_factory):
    """Destination partition with raw records: every rank may be handed ALL records; it keeps the
    in-edges of the nodes it owns (rank of the id in ascending order mod world)."""
    g = synth.RmatGraph(11, 12_000)
    e = g.edges(salt=1, salt_seed=9)
    fids, fvals, fst = hbo.faithful_run(e)
    world = 2
    flags = _lib.HB_FLAG_NO_RCCL | _lib.HB_FLAG_DEST_PARTITION
    ctxs = []
    try:
        for r in range(world):
            c = gpu_ctx_factory(rank=r, world_size=world, flags=flags)
            c.load_edges(e)
            c.begin()
            ctxs.append(c)
        has = True
        while has:
            for c in ctxs:
                c.step_local()
            _lib.Context.exchange(ctxs, 0)
            has = [c.step_finish() for c in ctxs][0]
        _lib.Context.exchange(ctxs, 1)
        for c in ctxs:
            c.finish()
            ids, vals = c.results()
            assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
        assert sum(c.stats()["m_eff"] for c in ctxs) == fst["m_eff"]
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("dest", [False, True, "changed", "edge_changed"])
def test_rccl_call_path_single_rank(gpu_ctx_factory, dest):
    """A 1-rank RCCL communicator: the collectives of both decompositions run for real -
    ncclAllReduce(max, u8) + epilogue (edge partition), grouped ncclAllGather of the counter / changed-bit
    slices + ncclAllReduce(sum) of the counters (destination partition), ncclAllGather of the Kahan sums."""
    g = synth.RmatGraph(12, 40_000)
    o, T, vals, keep, k = _oracle_dense(g.ids, g.row_ptr, g.src)
    uid = _lib.rccl_unique_id()
    flags = _lib.HB_FLAG_RCCL_SELF | (_lib.HB_FLAG_DEST_PARTITION if dest and dest != "edge_changed" else 0)
    flags |= _lib.HB_FLAG_CHANGED_ONLY if dest in ("changed", "edge_changed") else 0
    with gpu_ctx_factory(flags=flags, rccl_id=uid) as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        st = ctx.run()
        _check_final(ctx, g.ids, T, vals, keep, st)
        assert st["ms_collective"] > 0.0
        assert np.array_equal(ctx.registers(), o.registers())


# ---- larger sizes ---------------------------------------------------------------------------------
def test_c1_config_bit_exact(gpu_ctx_factory):
    cfg = synth.CONFIGS["C1"]
    g = synth.RmatGraph(cfg["scale"], cfg["m"])
    o, T, vals, keep, k = _oracle_dense(g.ids, g.row_ptr, g.src)
    with gpu_ctx_factory() as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        st = ctx.run()
        _check_final(ctx, g.ids, T, vals, keep, st)
        assert np.array_equal(ctx.registers(), o.registers())


def test_c2_config_bit_exact_and_properties(gpu_ctx_factory):
    """BASELINE.json configs[1]: 1M-host / 20M-edge, 1 GPU, bit-exact vs the oracle; plus
    size-independent properties: idempotence of a re-run, independence from the device
    layout (reorder / chunking / frontier), monotone counters."""
    cfg = synth.CONFIGS["C2"]
    g = synth.RmatGraph(cfg["scale"], cfg["m"])
    o, T, vals, keep, k = _oracle_dense(g.ids, g.row_ptr, g.src)
    outs = []
    for kw in (dict(), dict(flags=_lib.HB_FLAG_NO_REORDER | _lib.HB_FLAG_NO_FRONTIER, chunk=256)):
        with gpu_ctx_factory(**kw) as ctx:
            ctx.load_dense(g.ids, g.row_ptr, g.src)
            st = ctx.run()
            _check_final(ctx, g.ids, T, vals, keep, st)
            # [r5] at this size (n >= 2^20) the results travel while the passes run under the DEFAULT policy (first snapshot when the
            # frontier starts to shrink): on, and exact (the tail pipeline has its own tests: an R-MAT tail this short may end
            # with the first queued pass)
            assert g.n >= (1 << 20) and st["result_stages"] >= 1 and 0 < st["result_list"] < g.n // 8 + 4096, (g.n, st["result_stages"], st["result_list"])
            regs = ctx.registers()
            assert np.array_equal(regs, o.registers())
            st2 = ctx.run()  # re-running the loaded graph gives the same answer
            _check_final(ctx, g.ids, T, vals, keep, st2)
            outs.append(regs)
    assert np.array_equal(outs[0], outs[1])


def _mixed_page_host_documents(host_tuples, seed=11):
    """A page-level webgraph over the hosts of `host_tuples`: one document per host edge; the linking / linked page is
    the host's root page (page id == host id, webgraph/node.rs:140-156) or one of its sub-pages (another id)."""
    rng = np.random.default_rng(seed)
    host = graphs.EdgeListGraph.from_tuples(host_tuples).host_edges()
    page = host.copy()
    sub_from, sub_to = rng.random(len(host)) < 0.5, rng.random(len(host)) < 0.4
    page["from"]["lo"][sub_from] ^= rng.integers(1, 1 << 62, size=int(sub_from.sum()), dtype=np.uint64)
    page["to"]["lo"][sub_to] ^= rng.integers(1, 1 << 62, size=int(sub_to.sum()), dtype=np.uint64)
    return host, page


@pytest.mark.gpu
def test_reference_tail_mode(gpu_ctx_factory, tmp_path):
    """HB_FLAG_REFERENCE_TAIL: the reference's changed-node machinery as written (bloom filter with its false
    positives, exact-counting switch, sqrt(n) tail over page-level forward links) - against the
    faithful oracle given the same page-level records: ids, values (bits), pass count and the number of tail passes."""
    host = graphs.tailed_graph()
    foreign = [(0xDEAD0000 + k, host[k][1], 0) for k in range(5)] + [(host[k][0], 0xBEEF0000 + k, 0) for k in range(5)]
    flagged = [(f, t, graphs.NOFOLLOW) for f, t, _ in host[-40:]]
    e = graphs.EdgeListGraph.from_tuples(host).host_edges()
    to_edges = lambda tuples: graphs.EdgeListGraph.from_tuples(tuples).host_edges() if tuples else np.zeros(0, dtype=_lib.EDGE)
    cases = {"all_host_edges_as_pages": list(host), "no_call": None, "no_root_links": [], "every_other": host[::2] + foreign,
             "chain_links_flagged": host[:-40] + flagged}
    base_ids, base_vals, base_st = hbo.faithful_run(e)
    differs = 0
    for name, pages in cases.items():
        recs = None if pages is None else to_edges(pages)
        fids, fvals, fst = hbo.faithful_run(e, recs if recs is not None else np.zeros(0, dtype=hbo.EDGE))
        for extra in (0, _lib.HB_FLAG_HOST_PLAN):
            with gpu_ctx_factory(flags=_lib.HB_FLAG_REFERENCE_TAIL | extra) as ctx:
                ctx.load_edges(e)
                if recs is not None:
                    ctx.load_tail_edges(recs[::-1])   # (another document order)
                    ctx.load_tail_edges(recs)         # a second call replaces the first
                st = ctx.run()
                ids, vals = ctx.results()
                modes = [ps["mode"] for ps in ctx.pass_stats()]
            assert st["passes"] == fst["passes"], (name, modes)
            assert modes.count(3) == fst["passes_exact"], (name, modes)
            assert np.array_equal(ids, fids), name
            assert np.array_equal(vals.view(np.uint64), fvals.view(np.uint64)), name
        if name == "all_host_edges_as_pages":   # pages are hosts: same list as the default mode
            assert np.array_equal(ids, base_ids) and np.array_equal(vals.view(np.uint64), base_vals.view(np.uint64))
            assert fst["passes_exact"] > 10
        else:
            differs += int(len(vals) != len(base_vals) or not np.array_equal(vals.view(np.uint64), base_vals.view(np.uint64)))
    assert differs >= 3
    # a graph that falls back from the tail to update_all_counters with a stale bloom frontier: a wide levelled tail
    g = synth.RmatGraph(11, 9_000, tail=(600, 900, 3))
    e = g.edges(salt=1, salt_seed=5)
    rng = np.random.default_rng(3)
    recs = e[rng.random(len(e)) < 0.7]
    fids, fvals, fst = hbo.faithful_run(e, recs)
    with gpu_ctx_factory(flags=_lib.HB_FLAG_REFERENCE_TAIL) as ctx:
        ctx.load_edges(e)
        ctx.load_tail_edges(recs)
        st = ctx.run()
        ids, vals = ctx.results()
        modes = [ps["mode"] for ps in ctx.pass_stats()]
    assert st["passes"] == fst["passes"] and modes.count(3) == fst["passes_exact"], (modes, fst)
    assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
    # end to end from an on-disk store with page-level and host-level id columns (hb_load_webgraph + HBW_PAGE_IDS),
    # and the same records handed over in batches
    from stract_amd import webgraph
    from tests import tantivy_fixture as tf
    host_docs, page_docs = _mixed_page_host_documents(graphs.tailed_graph(chain=90))
    cut = [0, 500, 501, len(host_docs)]
    tf.write_edge_store(str(tmp_path / "edges"), [host_docs[a:b] for a, b in zip(cut, cut[1:])],
                        page_segments=[page_docs[a:b] for a, b in zip(cut, cut[1:])])
    fids, fvals, fst = hbo.faithful_run(host_docs, page_docs, [b - a for a, b in zip(cut, cut[1:])])
    hids, hvals, hst = hbo.faithful_run(host_docs)
    assert fst["passes"] < hst["passes"] and fst["passes_exact"] >= 1   # the page-level tail ends the run early here
    for how in ("store", "batches"):
        with gpu_ctx_factory(flags=_lib.HB_FLAG_REFERENCE_TAIL) as ctx:
            if how == "store":
                webgraph.load_webgraph(ctx, str(tmp_path / "edges"), verify_crc=True, page_ids=True)
            else:
                ctx.load_edges(host_docs)
                for part in np.array_split(page_docs, 5):
                    ctx.append_tail_edges(part)
            st = ctx.run()
            ids, vals = ctx.results()
            modes = [ps["mode"] for ps in ctx.pass_stats()]
        assert st["passes"] == fst["passes"] and modes.count(3) == fst["passes_exact"], (how, modes, fst)
        assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64)), how
    # duplicate page-level documents with conflicting rel flags: the query's LinksScorer keeps the first of a run of
    # neighbours per segment, harmonic.rs:87 filters on the survivor; a segment boundary in between resets it
    host = graphs.tailed_graph()
    e = graphs.EdgeListGraph.from_tuples(host).host_edges()
    pages = [(f, t, 0) for f, t, _ in host[:-60]]
    for k, (f, t, _) in enumerate(host[-60:]):
        pages += [(f, t, graphs.NOFOLLOW), (f, t, 0)] if k % 3 == 0 else [(f, t, 0), (f, t, graphs.NOFOLLOW)] if k % 3 == 1 else [(f, t, 0)]
    recs = to_edges(pages)
    cuts = [i + 1 for i, pg in enumerate(pages) if pg[2] and i + 1 < len(pages) and pages[i + 1][:2] == pg[:2]]
    results = []
    for segs in (None, [b - a for a, b in zip([0] + cuts, cuts + [len(pages)])]):
        fids, fvals, fst = hbo.faithful_run(e, recs, segs)
        with gpu_ctx_factory(flags=_lib.HB_FLAG_REFERENCE_TAIL) as ctx:
            ctx.load_edges(e)
            at = 0
            for cnt in (segs or [len(recs)]):
                for part in np.array_split(recs[at:at + cnt], 2):   # a segment may arrive in several batches
                    ctx.append_tail_edges(part)
                ctx.tail_segment_end()
                at += cnt
            st = ctx.run()
            ids, vals = ctx.results()
            modes = [ps["mode"] for ps in ctx.pass_stats()]
        assert st["passes"] == fst["passes"] and modes.count(3) == fst["passes_exact"], (modes, fst)
        assert np.array_equal(ids, fids) and np.array_equal(vals.view(np.uint64), fvals.view(np.uint64))
        results.append(vals.copy())
    assert len(results[0]) != len(results[1]) or not np.array_equal(results[0], results[1])   # the lost links matter
    # the mode is single-rank, and the records need the flag
    with pytest.raises(Exception):
        gpu_ctx_factory(flags=_lib.HB_FLAG_REFERENCE_TAIL | _lib.HB_FLAG_NO_RCCL, rank=0, world_size=2)
    with gpu_ctx_factory() as ctx:
        ctx.load_edges(e)
        with pytest.raises(Exception):
            ctx.load_tail_edges(recs)


@pytest.mark.gpu
def test_reference_two_shard_fixture(gpu_ctx_factory):
    """entrypoint/ampc/harmonic_centrality/mod.rs:90-184 (test_simple_graph): the fixture's edges dealt i % 2 over two
    shards, the distributed result compared with HarmonicCentrality::calculate on the combined graph (there: 1e-4; here
    bit for bit).  Edge partition over two logical ranks, raw records, the all-reduce emulated on one device."""
    edges = graphs.fixture_graph().host_edges()
    want = HarmonicCentrality.calculate(graphs.fixture_graph())
    ids = np.unique(np.concatenate([edges["from"], edges["to"]]))
    ids = ids[np.lexsort((ids["lo"], ids["hi"]))]
    ctxs = []
    try:
        for r in range(2):
            c = gpu_ctx_factory(rank=r, world_size=2, flags=_lib.HB_FLAG_NO_RCCL)
            c.load_edges(edges[r::2], ids)          # every shard knows all nodes, holds its own edges
            c.begin()
            ctxs.append(c)
        has = True
        while has:
            for c in ctxs:
                c.step_local()
            _lib.Context.exchange(ctxs, 0)
            has = [c.step_finish() for c in ctxs][0]
        _lib.Context.exchange(ctxs, 1)
        for c in ctxs:
            c.finish()
            got_ids, got_vals = c.results()
            assert np.array_equal(got_ids, want.arrays()[0])
            assert np.array_equal(got_vals.view(np.uint64), want.arrays()[1].view(np.uint64))
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,modes", [(2, "edge,edge_no_pipeline,edge_changed,dest,dest_changed"), (3, "edge,dest_changed")])
def test_multi_process_library_exchanges_on_one_device(tmp_path, world, modes):
    """N PROCESSES (torch.distributed.run, gloo) drive the library's multi-rank pass driver on ONE device; the
    exchanges go through hb_set_collectives + stract_amd.dist.HostStagedCollectives (host-staged gloo tensors) in place of RCCL,
    which refuses two ranks on one device.  Every decomposition must give the single-GPU result bit for bit; the counters
    must be identical on all ranks; the callbacks must have carried the traffic (call counts)."""
    import json
    import socket
    import subprocess
    import sys

    scale, m = 12, 40_000
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dist_gpu_worker.py"), str(scale), str(m), str(tmp_path), modes]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = synth.RmatGraph(scale, m)
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    T = o.run()
    ovals, keep, k = o.finish()
    want_bits = ovals[keep].view(np.uint64).tolist()
    ranks = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % i))) for i in range(world)]
    for mode in modes.split(","):
        edges = 0
        for i, z in enumerate(ranks):
            z = z[mode]
            assert z["callback_error"] is None, (mode, i, z["callback_error"])
            assert z["passes"] == T and z["results"] == k, (mode, i, z["passes"], T)
            assert z["same_counters_on_all_ranks"], (mode, i)
            assert z["calls"]["all_reduce"] + z["calls"]["all_gather"] + z["calls"]["broadcast"] >= T, (mode, z["calls"])
            edges += z["local_edges"]
        assert edges == g.m, mode
        assert ranks[0][mode]["vals_bits"] == want_bits, mode          # the final list, every f64 bit
        if mode == "edge":   # the pipelined form: four all-reduces (row ranges) per pass + the out-degree histogram at load + Kahan slices
            assert ranks[0][mode]["calls"]["all_reduce"] >= 4 * T, ranks[0][mode]["calls"]
        if mode == "edge_no_pipeline":
            assert T <= ranks[0][mode]["calls"]["all_reduce"] <= T + 2, ranks[0][mode]["calls"]
        if mode.endswith("changed"):
            assert ranks[0][mode]["calls"]["all_gather"] >= T, ranks[0][mode]["calls"]


@pytest.mark.gpu
def test_caching_allocator_under_memory_pressure(tmp_path):
    """hb_pool.h keeps freed device extents for reuse while the device has room and gives blocks back to the runtime once what it
    holds passes its limit.  A child process with HB_POOL_LIMIT_BYTES = 64 MiB (every larger request trims the cache first - the
    regime of the 5 B-edge graph on a full device) must give the same graph, plan and result as this process, whose limit is
    half of the device; and its pool high-water mark must stay near what was live."""
    import subprocess
    import sys
    code = (
        "import json, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from stract_amd import _lib, synth\n"
        "cfg = synth.CONFIGS['C2']; g = synth.RmatGraph(cfg['scale'], cfg['m'])\n"
        "recs = np.zeros(g.stream_len(2), dtype=_lib.EDGE); g.stream_fill(recs, 0, 2)\n"
        "out = []\n"
        "for rnd in range(2):\n"
        "    with _lib.Context() as ctx:\n"
        "        for part in np.array_split(recs, 7): ctx.append_edges(part)\n"
        "        ctx.finalize(); st = ctx.run(); ids, vals = ctx.results(); hr, hk = ctx.state_hash()\n"
        "    out.append([int(st['n']), int(st['m_eff']), int(st['passes']), int(len(vals)), int(vals.view(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF), hr, hk,\n"
        "                int(st['ingest_peak_bytes']), int(st['pool_peak_bytes'])])\n"
        "print(json.dumps(out))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    runs = {}
    for name, env in (("roomy", {}), ("tight", {"HB_POOL_LIMIT_BYTES": str(64 << 20)})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for a, b in zip(runs["roomy"], runs["tight"]):
        assert a[:7] == b[:7], (a, b)                      # same graph, passes, results, state checksums
    assert runs["roomy"][0][:7] == runs["roomy"][1][:7]   # a second load out of the cached extents
    live, held_roomy, held_tight = runs["tight"][0][7], runs["roomy"][0][8], runs["tight"][0][8]
    assert live > 0 and held_tight >= live * 0 and held_tight <= held_roomy, (live, held_roomy, held_tight)
