"""CPU oracle (oracle/hb_oracle.c) against every known answer the reference's tests hold
for this path, SURVEY.md Appendix B, and the committed golden fixtures."""
import ctypes
import json
import math
import os

import numpy as np
import pytest

from oracle import hbo, pyref
from tests import graphs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hyperball_golden.json")


def test_kahan_known_answer():
    # kahan_sum.rs:87-125 - the only exact float the reference pins on this path
    vals = [10000.0, math.pi, math.e, math.pi, math.e, math.pi, math.e]
    s, _ = hbo.kahan_sum(vals)
    assert s == 10017.579623446147


def test_hll_add_positions():
    # add(1) sets reg[39] = 1, add(0) sets reg[0] = 65
    r = np.zeros(64, np.uint8)
    hbo.hll_add(r, 1)
    assert r[39] == 1 and r.sum() == 1
    r[:] = 0
    hbo.hll_add(r, 0)
    assert r[0] == 65 and np.count_nonzero(r) == 1
    # only the low 64 bits of a u128 id matter (hyperloglog.rs:4398-4400)
    r2 = np.zeros(64, np.uint8)
    hbo.hll_add(r2, (123 << 64) | 0)
    assert np.array_equal(r, r2)


def test_hll_size_sequence():
    # SURVEY.md Appendix B (independent survey-time derivation)
    ks = [1, 2, 3, 4, 5, 10, 20, 35, 50, 64, 100, 128, 200, 500, 1000, 10**4, 10**5, 10**6]
    exp = [1, 2, 3, 4, 5, 10, 23, 67, 89, 110, 181, 226, 364, 875, 1782, 16753, 184912, 1794420]
    r = np.zeros(64, np.uint8)
    got, j = [], 0
    for i in range(10**6):
        hbo.hll_add(r, i)
        if i + 1 == ks[j]:
            got.append(hbo.hll_size(r))
            j += 1
    assert got == exp


def test_hll_merge_is_union():
    # hyperloglog.rs:4579-4598 `merge` (there with N = 128): registers of a merged pair
    # equal those of a counter that saw both streams
    L = hbo.load()
    whole, a, b = (np.zeros(64, np.uint8) for _ in range(3))
    for i in range(10_000):
        hbo.hll_add(whole, i)
        hbo.hll_add(a, i)
    for i in range(10_001, 20_000):
        hbo.hll_add(whole, i)
        hbo.hll_add(b, i)
    L.hbo_hll_merge(a.ctypes.data, b.ctypes.data)
    assert np.array_equal(a, whole)


# hyperloglog.rs:4553-4611 (size_estimate_within_bounds, many_different_sizes,
# accurate_counts) instantiate N = 128 / 65536 only; with N = 64 the estimator indexes the
# precision-5 bias rows (SURVEY.md surprise 3) and sequential items under FastHasher give
# e.g. size(0..10^4) = 16753, so those accuracy bounds do not apply to this path and are
# not asserted - test_hll_size_sequence pins the exact N = 64 values instead.


def _run_faithful(graph):
    ids, vals, st = hbo.faithful_run(graph.host_edges())
    return {(int(h) << 64) | int(l): v for l, h, v in zip(ids["lo"], ids["hi"], vals)}, st


def test_reference_fixture_ordering_and_values():
    # harmonic.rs:460-474: C > A > B and D absent
    res, st = _run_faithful(graphs.fixture_graph())
    A, B, C, D = graphs.A, graphs.B, graphs.C, graphs.D
    assert res[C] > res[A] > res[B]
    assert D not in res
    # SURVEY.md Appendix B hand derivation: ball sizes A:1,2,4  B:1,2,3,4  C:1,4
    assert st["passes"] == 4
    assert np.float64(res[A]).view(np.uint64) == 0x3FE5555555555555
    assert np.float64(res[B]).view(np.uint64) == 0x3FE38E38E38E38E3
    assert np.float64(res[C]).view(np.uint64) == 0x3FF0000000000000


def test_reference_host_fixture():
    # harmonic.rs:358-458: B.com > A.com (A.com has only a self-loop -> absent -> 0.0)
    g, (a, b, c, d) = graphs.host_fixture()
    res, _ = _run_faithful(g)
    assert res[b] > res.get(a, 0.0)
    assert a not in res


def test_additional_edges_ignored():
    # harmonic.rs:476-528: duplicate edges give an identical map
    base, _ = _run_faithful(graphs.fixture_graph())
    extra, _ = _run_faithful(graphs.fixture_graph(extra=[(graphs.A, graphs.B, 0)] * 8))
    assert base == extra


@pytest.mark.parametrize("flag", [graphs.TAG, graphs.SAME_ICANN_DOMAIN])
def test_rel_flags_ignored(flag):
    # harmonic.rs:530-578: every value == 0.0, i.e. nothing survives the > 0 filter
    res, st = _run_faithful(graphs.fixture_graph(flags=flag))
    assert res == {}
    assert st["n"] == 4 and st["m_eff"] == 0


def test_first_occurrence_flag_wins():
    # store.rs:313 (unique_by) precedes harmonic.rs:131 (flag filter): a later clean copy
    # of a flagged first record stays lost, a later flagged copy of a clean record is ignored
    A, B, C = 10, 20, 30
    lost, _ = _run_faithful(graphs.EdgeListGraph.from_tuples([(A, B, graphs.NOFOLLOW), (A, B, 0), (B, C, 0)]))
    assert B not in lost and C in lost
    kept, _ = _run_faithful(graphs.EdgeListGraph.from_tuples([(A, B, 0), (A, B, graphs.NOFOLLOW), (B, C, 0)]))
    assert B in kept and C in kept


def test_lcg_graph_appendix_b():
    # 7 passes, 200 results, first five values
    edges = graphs.lcg_graph()
    ids, row_ptr, src = graphs.dense_from_tuples(edges)
    o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
    T = o.run()
    vals, keep, k = o.finish()
    assert T == 7 and k == 200
    exp = [0.5506700167504188, 0.6469849246231156, 0.612646566164154, 0.6017587939698492, 0.5182579564489113]
    assert vals[:5].tolist() == exp


def test_dense_equals_faithful_on_salted_rmat():
    from stract_amd import _lib, synth

    g = synth.RmatGraph(11, 12_000)
    e = g.edges(salt=1, salt_seed=7)
    fids, fvals, fst = hbo.faithful_run(e)
    ids, row_ptr, src, mu = _lib.host_ingest(e)
    assert fst["n"] == len(ids) and fst["m_unique"] == mu and fst["m_eff"] == len(src)
    o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
    for flags in (0, hbo.FRONTIER, hbo.FRONTIER | hbo.LITERAL):
        o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
        T = o.run(flags)
        vals, keep, k = o.finish()
        assert T == fst["passes"]
        assert np.array_equal(ids[keep], fids)
        assert np.array_equal(vals[keep].view(np.uint64), fvals.view(np.uint64))
    assert fst["passes_exact"] > 0  # the sqrt(n) tail mode was exercised


def test_threads_do_not_change_results():
    from stract_amd import synth

    g = synth.RmatGraph(12, 40_000)
    outs = []
    for th in (1, 4):
        o = hbo.Dense(g.id_low64(), g.row_ptr, g.src, threads=th)
        o.run()
        outs.append((o.registers(), o.kahan()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1][0].view(np.uint64), outs[1][1][0].view(np.uint64))


def test_binary_search_variants_agree_on_reachable_estimates():
    # The raw-estimate table is unsorted at indices 127/128 and 130/131.
    # Scan e densely over the table range + every table value and its neighbours.
    L = hbo.load()
    raw = np.array([L.hbo_hll_bias_first_index(0.0, 0)])  # noqa: F841 (load check)
    es = list(np.linspace(20.0, 330.0, 200_001))
    diffs = 0
    for e in es:
        if L.hbo_hll_estimate_bias(e, 0) != L.hbo_hll_estimate_bias(e, 1):
            diffs += 1
    # the two std versions may pick different first neighbours only inside the two unsorted
    # spots; record how often (golden) rather than assume zero
    gold = json.load(open(GOLD))
    assert diffs == gold["bsearch_variant_disagreements_linspace_20_330_200001"]


def test_golden_vectors():
    gold = json.load(open(GOLD))
    regs = np.array(gold["size_cases"]["registers"], dtype=np.uint8)
    assert hbo.hll_sizes(regs).tolist() == gold["size_cases"]["sizes"]
    for case in gold["graphs"]:
        ids, row_ptr, src = graphs.dense_from_tuples([tuple(e) for e in case["edges"]])
        o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
        T = o.run()
        vals, keep, k = o.finish()
        assert T == case["passes"]
        got = {str((int(h) << 64) | int(l)): float(v).hex() for l, h, v in zip(ids["lo"][keep], ids["hi"][keep], vals[keep])}
        assert got == case["centrality_hex"]


def test_rank_results_order():
    # centrality/mod.rs:92-103 + lib.rs:259-263: (Reverse(total_cmp(centrality)), NodeID ascending)
    vals = np.array([0.5, 0.75, 0.5, 0.0, 1.0, 0.75, 5e-324], dtype=np.float64)
    r = hbo.rank_results(vals)
    assert r.tolist() == [3, 1, 4, 6, 0, 2, 5]
    rng = np.random.default_rng(3)
    v = rng.choice(rng.random(50), 2000)  # many ties
    want = np.empty(len(v), dtype=np.uint64)
    want[np.lexsort((np.arange(len(v)), -v))] = np.arange(len(v), dtype=np.uint64)
    assert np.array_equal(hbo.rank_results(v), want)
    assert len(hbo.rank_results(np.zeros(0))) == 0
    # sorted_k's own test data (centrality/mod.rs:120-205 pins a top-k ordering by value only)


def test_second_restatement_agrees_estimator():
    """oracle/pyref.py (plain Python, written from the Rust sources independently of hb_oracle.c) against the C
    oracle: HyperLogLog add positions and size() on every branch."""
    from oracle import pyref

    r = pyref.hll_new()
    pyref.hll_add(r, 1)
    assert r[39] == 1 and sum(r) == 1
    r = pyref.hll_new()
    pyref.hll_add(r, (123 << 64) | 0)  # add_u128 truncates to the low 64 bits
    assert r[0] == 65
    gold = json.load(open(GOLD))
    regs = np.array(gold["size_cases"]["registers"], dtype=np.uint8)
    assert [pyref.hll_size(x.tolist()) for x in regs] == gold["size_cases"]["sizes"]
    rng = np.random.default_rng(11)
    more = graphs.random_registers(rng, 3000)
    want = hbo.hll_sizes(more).tolist()
    assert [pyref.hll_size(x.tolist()) for x in more] == want
    # counters as they occur in a run
    c = pyref.hll_new()
    cc = np.zeros(64, np.uint8)
    for i, x in enumerate(rng.integers(0, 1 << 63, size=4000, dtype=np.uint64)):
        pyref.hll_add(c, int(x))
        hbo.hll_add(cc, int(x))
        if i % 97 == 0:
            assert c == cc.tolist()
            assert pyref.hll_size(c) == hbo.hll_size(cc)
    # the unsorted spots of the raw-estimate table: both binary searches land on the same index
    L = hbo.load()
    for e in list(np.linspace(127.0, 133.0, 4001)) + pyref.RAW:
        kind, i = pyref.binary_search_by(pyref.RAW, float(e))
        first = len(pyref.RAW) - 1 if (kind == "Err" and i == len(pyref.RAW)) else i
        assert first == L.hbo_hll_bias_first_index(ctypes.c_double(float(e)), 0)


def test_second_restatement_agrees_hyperball():
    """The map-based Python HyperBall against the C oracle (faithful form), bit for bit."""
    from oracle import pyref

    rng = np.random.default_rng(5)
    cases = [[(f, t, 0) for f, t in graphs.FIXTURE],
             [(f, t, 0) for f, t in graphs.lcg_graph(50, 120, 99)],
             [(f, t, graphs.NOFOLLOW if i % 7 == 0 else 0) for i, (f, t) in enumerate(graphs.lcg_graph(80, 300, 5))],
             [(1, 2, graphs.TAG), (1, 2, 0), (2, 3, 0), (3, 3, 0), (9, 1, graphs.SAME_ICANN_DOMAIN)]]
    for _ in range(6):
        kind, edges = graphs.random_graph(rng)
        cases.append([(f, t, 0) for f, t in edges[:1500]])
    for tuples in cases:
        py, passes = pyref.harmonic_centrality(tuples)
        ids, vals, st = hbo.faithful_run(graphs.EdgeListGraph.from_tuples(tuples).host_edges())
        got = {(int(h) << 64) | int(l): float(v) for l, h, v in zip(ids["lo"], ids["hi"], vals)}
        assert st["passes"] == passes
        assert list(got.keys()) == list(py.keys())
        assert [np.float64(v).view(np.uint64) for v in got.values()] == [np.float64(v).view(np.uint64) for v in py.values()]


def test_bloom_pieces():
    # bloom/src/lib.rs:36-41: bits = ceil(n ln(0.05) / (-8 ln^2 2)) ~= 0.78 n
    assert hbo.load().hbo_bloom_num_bits(1000, 0.05) == 780
    L = hbo.load()
    # :108-123 - the logarithm is truncated to i64 BEFORE the multiplication
    assert L.hbo_bloom_estimate_card(1000, 0) == 0
    assert L.hbo_bloom_estimate_card(1000, 600) == 0        # ln(0.4) = -0.91 -> 0
    assert L.hbo_bloom_estimate_card(1000, 700) == 1000     # ln(0.3) = -1.2  -> -1
    assert L.hbo_bloom_estimate_card(1000, 1000) == 0xFFFFFFFFFFFFFFFF


def test_oracle_under_sanitizers(tmp_path):
    """The C oracle rebuilt with AddressSanitizer + UBSan, exercised through the same Python wrapper on the
    golden vectors and two HyperBall runs (sanitizer runs of the CPU oracle)."""
    import shutil
    import subprocess
    import sys
    import textwrap

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / "san"
    shutil.copytree(os.path.join(root, "oracle"), work / "oracle", ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    shutil.copytree(os.path.join(root, "stract_amd"), work / "stract_amd", ignore=shutil.ignore_patterns("csrc", "__pycache__"))
    shutil.copytree(os.path.join(root, "tests"), work / "tests", ignore=shutil.ignore_patterns("__pycache__"))
    r = subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-std=c11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-ffp-contract=off", "-fopenmp", "-D_POSIX_C_SOURCE=200809L", "-shared", "-o",
                        str(work / "oracle" / "libhb_oracle.so"), str(work / "oracle" / "hb_oracle.c"), "-lm"],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available")
    assert r.returncode == 0, r.stderr[-2000:]
    pre = [subprocess.run(["gcc", "-print-file-name=" + lib], capture_output=True, text=True).stdout.strip()
           for lib in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(p) and os.path.exists(p) for p in pre):
        pytest.skip("sanitizer runtime libraries not found")
    script = textwrap.dedent("""
        import json, numpy as np
        from oracle import hbo
        from tests import graphs
        from stract_amd import synth
        gold = json.load(open("tests/golden/hyperball_golden.json"))
        regs = np.array(gold["size_cases"]["registers"], dtype=np.uint8)
        assert hbo.hll_sizes(regs).tolist() == gold["size_cases"]["sizes"]
        ids, row_ptr, src = graphs.dense_from_tuples(graphs.lcg_graph())
        o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src, threads=2)
        assert o.run() == 7
        g = synth.RmatGraph(10, 6000, threads=2)
        fids, fvals, st = hbo.faithful_run(g.edges(salt=1, salt_seed=3))
        assert st["passes"] > 2 and len(fvals) > 0
        assert hbo.rank_results(fvals).max() == len(fvals) - 1
        print("sanitized oracle ok")
    """)
    env = dict(os.environ, LD_PRELOAD=" ".join(pre), ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2", PYTHONPATH=str(work))
    r = subprocess.run([sys.executable, "-c", script], cwd=str(work), capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_state_hash_definition():
    """hbo_dense_state_hash (the checksum bench.py compares per pass at sizes too big to ship) against a
    plain-Python statement of its definition."""
    M = (1 << 64) - 1

    def mix(x):
        x ^= x >> 33
        x = (x * 0xff51afd7ed558ccd) & M
        x ^= x >> 33
        x = (x * 0xc4ceb9fe1a85ec53) & M
        x ^= x >> 33
        return x

    from stract_amd import synth

    g = synth.RmatGraph(9, 3000)
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    for _ in range(3):
        o.step(hbo.FRONTIER)
    regs = o.registers()
    ks, ke = o.kahan()
    hr = hk = 0
    for v in range(g.n):
        r = (v * 0x9E3779B97F4A7C15 + 1) & M
        for w in regs[v].view("<u8"):
            r = mix(r ^ int(w))
        hr = (hr + r) & M
        a, b = int(ks[v:v + 1].view(np.uint64)[0]), int(ke[v:v + 1].view(np.uint64)[0])
        hk = (hk + mix(mix(((v + 0x632BE59BD9B4E019) & M) ^ a) ^ b)) & M
    assert o.state_hash() == (hr, hk)
    o2 = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    o2.step(hbo.FRONTIER)
    assert o2.state_hash() != (hr, hk)


# ---- the reference's own HyperLogLog tests (hyperloglog.rs:4553-4611), through the generalised restatement ---------
def _np_registers(n, items):
    """Registers after adding `items` (numpy u64) to a fresh HyperLogLog<n> with FastHasher - vectorised
    hyperloglog.rs:4385-4396; checked against pyref.hll_add on a sample."""
    b = pyref.hll_b(n)
    with np.errstate(over="ignore"):
        h = items.astype(np.uint64) * np.uint64(11400714819323198549)
    j = (h >> np.uint64(64 - b)).astype(np.int64)
    w = h << np.uint64(b)
    lz = np.full(len(w), 64, dtype=np.int64)  # leading_zeros of a u64 (64 for 0)
    x = w.copy()
    nz = x != 0
    lz[nz] = 0
    for shift in (32, 16, 8, 4, 2, 1):
        top = (x >> np.uint64(64 - shift)) == 0
        m = nz & top
        lz[m] += shift
        x[m] = x[m] << np.uint64(shift)
    reg = np.zeros(n, dtype=np.int64)
    np.maximum.at(reg, j, lz + 1)
    return [int(v) for v in reg]


def test_reference_hll_size_estimate_within_bounds():
    # hyperloglog.rs:4554-4564 (10 M items) and :4566-4576 (10 k items), HyperLogLog<128>
    for count in (10_000_000, 10_000):
        reg = _np_registers(128, np.arange(count, dtype=np.uint64))
        if count == 10_000:  # the vectorised add equals the line-by-line one
            slow = pyref.hll_new(128)
            for item in range(count):
                pyref.hll_add(slow, item)
            assert slow == reg
        size = pyref.hll_size(reg)
        lo, hi = pyref.hll_size_bounds(reg)
        assert lo < size < hi  # all the reference asserts (its FastHasher on sequential ids is far off at 10 M: 18.5 M)


def test_reference_hll_merge():
    # hyperloglog.rs:4578-4598
    without_merge = _np_registers(128, np.concatenate([np.arange(10_000), np.arange(10_001, 20_000)]).astype(np.uint64))
    a = _np_registers(128, np.arange(10_000, dtype=np.uint64))
    b = _np_registers(128, np.arange(10_001, 20_000, dtype=np.uint64))
    pyref.hll_merge(a, b)
    assert a == without_merge


def test_reference_hll_accurate_counts():
    # hyperloglog.rs:4600-4611: HyperLogLog<65_536>, after each of 1000 adds |size - count| <= 10
    reg = pyref.hll_new(65_536)
    for counter, item in enumerate(range(1_000), start=1):
        pyref.hll_add(reg, item)
        assert abs(pyref.hll_size(reg) - counter) <= 10.0, counter


def test_generalised_estimator_agrees_with_the_c_oracle_at_64():
    rng = np.random.default_rng(11)
    for regs in graphs.random_registers(rng, 300):
        assert pyref.hll_size([int(x) for x in regs]) == hbo.hll_size(regs)


def _tuples_to_edges(tuples):
    return graphs.EdgeListGraph.from_tuples(tuples).host_edges() if tuples else np.zeros(0, dtype=hbo.EDGE)


def _faithful_dict(edges, pages, segments=None):
    ids, vals, st = hbo.faithful_run(edges, pages, segments)
    return {(int(h) << 64) | int(l): float(v) for l, h, v in zip(ids["lo"], ids["hi"], vals)}, st


def test_reference_tail_two_restatements_agree():
    """The reference's loop AS WRITTEN (bloom filter with false positives, exact-counting switch, sqrt(n) tail over
    page-level forward links - SURVEY.md App. C-5): the Python restatement against the C oracle, bit for bit."""
    from oracle import pyref

    host = graphs.tailed_graph()
    foreign = [(0xDEAD0000 + k, host[k][1], 0) for k in range(5)] + [(host[k][0], 0xBEEF0000 + k, 0) for k in range(5)]
    flagged = [(f, t, graphs.NOFOLLOW) for f, t, _ in host[-40:]]
    cases = {
        "host_level": None,                          # pages are hosts: the default semantics
        "all_host_edges_as_pages": list(host),       # same thing said through the page-level interface
        "no_root_links": [],                         # the query finds nothing: the tail stops at once
        "every_other": host[::2] + foreign,          # some links exist at page level; unknown ids fall out (harmonic.rs:91-92)
        "chain_links_flagged": host[:-40] + flagged, # rel filter on the query result (harmonic.rs:87)
    }
    e = _tuples_to_edges(host)
    base, base_passes = pyref.harmonic_centrality(host)
    seen = {}
    for name, pages in cases.items():
        py, passes, tail_passes = pyref.harmonic_centrality_reference(host, pages)
        got, st = _faithful_dict(e, None if pages is None else _tuples_to_edges(pages))
        assert (st["passes"], st["passes_exact"]) == (passes, tail_passes), name
        assert list(got.keys()) == list(py.keys()), name
        assert [np.float64(v).view(np.uint64) for v in got.values()] == [np.float64(v).view(np.uint64) for v in py.values()], name
        seen[name] = (py, passes, tail_passes)
    # host-level pages: the machinery is results-inert (App. C-1) - same list as the plain iteration
    for name in ("host_level", "all_host_edges_as_pages"):
        assert seen[name][0] == base and seen[name][1] == base_passes and seen[name][2] > 0
    # page-level tail: the run ends early and the end of the chain keeps smaller values (documented in DESIGN.md §5)
    py, passes, tail_passes = seen["no_root_links"]
    assert passes < base_passes and tail_passes == 1
    diff = [k for k in base if base[k] != py.get(k)]
    assert 0 < len(diff) < 60 and all(py.get(k, 0.0) < base[k] for k in diff)


def test_links_scorer_two_restatements_agree():
    """LinksScorer (query/raw/links.rs:115-232), the de-duplication a ForwardlinksQuery applies BEFORE harmonic.rs:87
    sees any flags: pyref simulates tantivy's block cursor literally, the C oracle states the same walk in closed form.
    Hand cases first (what the code does, as written), then random posting lists across the 128-document block sizes."""
    import random

    from oracle import pyref

    def c_emit(vals, me):
        to = np.zeros(len(vals), dtype=hbo.U128)
        to["lo"] = np.array(vals, dtype=np.uint64)
        meid = np.zeros(1, dtype=hbo.U128)
        meid["lo"] = me
        return [int(i) for i in np.nonzero(hbo.links_scorer(to, meid[0]))[0]]

    hand = [
        ([], 9, []),
        ([9, 9, 9], 9, []),                               # only self links
        ([9, 2, 2, 3, 2], 9, [1, 3, 4]),                  # adjacent duplicates only: the second `2` run is yielded again
        ([2, 9, 2, 3], 9, [0, 3]),                        # a self link between two equal targets does not reset the memory
        ([2] + [3] * 126 + [2] + [4, 5], 9, [0, 128, 129]),  # block jump: the LAST doc of the full block repeats -> `3` is never seen
        ([2] + [3] * 126 + [4] + [4, 5], 9, [0, 1, 127, 129]),
        ([1, 2] + [3] * 125 + [2] + [4] * 127 + [2] + [5], 9, [0, 1, 256]),
    ]
    for vals, me, want in hand:
        assert pyref.links_scorer_docs(vals, me) == want, vals[:8]
        assert c_emit(vals, me) == want, vals[:8]
    rng = random.Random(5)
    for _ in range(1500):
        n = rng.choice([1, 2, 5, 127, 128, 129, 255, 256, 257, 300, 513])
        me = 7
        vals = [rng.choice([1, 2, 3, me]) if rng.random() < 0.7 else rng.randrange(1, 6) for _ in range(n)]
        if rng.random() < 0.4:
            for b in range(0, n - 127, 128):
                if rng.random() < 0.6:
                    vals[b + 127] = vals[b - 1] if b else vals[0]
        assert c_emit(vals, me) == pyref.links_scorer_docs(vals, me)


def test_reference_tail_conflicting_duplicate_flags():
    """Page-level duplicates of one (from_id, to_id) with DIFFERENT rel flags: the query keeps the first of a run of
    neighbours (per segment), harmonic.rs:87 then filters on that survivor's flags only - so a flagged document in front
    of a clean one loses the link, the other order keeps it, and a segment boundary between them keeps it too."""
    from oracle import pyref

    host = graphs.tailed_graph()
    e = _tuples_to_edges(host)
    chain = host[-60:]
    pages, lost, kept = [], set(), set()
    for k, (f, t, _) in enumerate(host[:-60]):
        pages.append((f, t, 0))
    for k, (f, t, _) in enumerate(chain):
        if k % 3 == 0:
            pages += [(f, t, graphs.NOFOLLOW), (f, t, 0)]     # flagged first: the clean copy is de-duplicated away
            lost.add((f, t))
        elif k % 3 == 1:
            pages += [(f, t, 0), (f, t, graphs.NOFOLLOW)]     # clean first: kept
            kept.add((f, t))
        else:
            pages.append((f, t, 0))
    nodes = {x for a, b, _ in host for x in (a, b)}
    fwd = pyref.forwardlinks_result(pages, nodes)
    got = {(f, t) for f, ts in fwd.items() for t in ts}
    assert not (got & lost) and kept <= got
    naive = {(f, t) for f, t, fl in pages if not fl & 0x6FED00}
    assert lost <= naive                                      # "any duplicate passes" would keep them: the divergence
    # one segment, and the same documents with a boundary right after every flagged-first document
    py1, p1, t1 = pyref.harmonic_centrality_reference(host, pages)
    g1, st1 = _faithful_dict(e, _tuples_to_edges(pages))
    assert (st1["passes"], st1["passes_exact"]) == (p1, t1) and list(g1.keys()) == list(py1.keys())
    assert [np.float64(v).view(np.uint64) for v in g1.values()] == [np.float64(v).view(np.uint64) for v in py1.values()]
    cuts = [i + 1 for i, pg in enumerate(pages) if pg[2] and (pg[0], pg[1]) in lost]
    segs = [b - a for a, b in zip([0] + cuts, cuts + [len(pages)])]
    py2, p2, t2 = pyref.harmonic_centrality_reference(host, pages, segs)
    g2, st2 = _faithful_dict(e, _tuples_to_edges(pages), segs)
    assert (st2["passes"], st2["passes_exact"]) == (p2, t2) and list(g2.keys()) == list(py2.keys())
    assert [np.float64(v).view(np.uint64) for v in g2.values()] == [np.float64(v).view(np.uint64) for v in py2.values()]
    base, _ = pyref.harmonic_centrality(host)
    assert py2 == base or p2 >= p1                            # with the boundaries every link is found again
    assert py1 != py2                                         # the lost links change the result


def test_pyref_bloom_matches_c_pieces():
    from oracle import pyref

    L = hbo.load()
    for n in (1, 2, 7, 100, 1000, 12345, 10 ** 6):
        assert pyref.bloom_num_bits(n) == L.hbo_bloom_num_bits(n, 0.05)
    b = pyref.Bloom(1000)
    for ones in (0, 1, 300, 492, 493, 600, 700, 779):
        b.bits = set(range(ones))
        assert b.estimate_card() == L.hbo_bloom_estimate_card(b.num_bits, ones), ones


def test_reference_bloom_filter_test():
    """crates/bloom/src/lib.rs:199-216 (test_bloom_filter) on the restated U64BloomFilter: U64BloomFilter::new(100, 0.01),
    1..5 inserted -> contained, 6..10 not.  Holds only with the reference's bit count and multiplicative hash."""
    from oracle import pyref

    bf = pyref.Bloom(100, 0.01)
    assert bf.num_bits == hbo.load().hbo_bloom_num_bits(100, 0.01) == 120
    for v in (1, 2, 3, 4, 5):
        bf.insert(v)
    assert all(bf.contains(v) for v in (1, 2, 3, 4, 5))
    assert not any(bf.contains(v) for v in (6, 7, 8, 9, 10))
