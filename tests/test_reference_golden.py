"""Golden vectors produced by the REFERENCE itself (tools/ref_golden.rs, to be run on a machine with cargo).  While
none are committed the oracle stays "parity unpinned" (oracle/hb_oracle.h) and these tests skip; the day
tests/golden/reference_*.json appear, the oracle (both forms), the HIP path and - with tests/golden/reference_store -
the native column reader are compared with them, no other change needed."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import hbo
from stract_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "reference_*.json")))
STORE = os.path.join(HERE, "golden", "reference_store", "edges")


def load_case(path):
    d = json.load(open(path))
    e = np.zeros(len(d["edges"]), dtype=_lib.EDGE)
    for i, (f, t, flags) in enumerate(d["edges"]):
        f, t = int(f, 16), int(t, 16)
        e[i]["from"] = (f & 0xFFFFFFFFFFFFFFFF, f >> 64)
        e[i]["to"] = (t & 0xFFFFFFFFFFFFFFFF, t >> 64)
        e[i]["rel_flags"] = flags
    want = [(int(i, 16), int(b, 16)) for i, b in d["centrality"]]
    return d["name"], e, want


def load_pages(path):
    """Page-level records of a mixed page/host case ("pages", tools/ref_golden.rs graph 4), or None."""
    d = json.load(open(path))
    if "pages" not in d:
        return None
    p = np.zeros(len(d["pages"]), dtype=_lib.EDGE)
    for i, (f, t, flags) in enumerate(d["pages"]):
        f, t = int(f, 16), int(t, 16)
        p[i]["from"] = (f & 0xFFFFFFFFFFFFFFFF, f >> 64)
        p[i]["to"] = (t & 0xFFFFFFFFFFFFFFFF, t >> 64)
        p[i]["rel_flags"] = flags
    return p


def as_pairs(ids, vals):
    return [((int(h) << 64) | int(l), int(b)) for l, h, b in zip(ids["lo"], ids["hi"], vals.view(np.uint64))]


def test_fixture_loader_round_trip(tmp_path):
    """The loader itself is exercised even without reference files: a file in the harness' format, written from
    the oracle, must read back to the same records and values."""
    from stract_amd import synth
    g = synth.RmatGraph(8, 400)
    e = g.edges(salt=1, salt_seed=2)
    ids, vals, _ = hbo.faithful_run(e)
    u = lambda r: (int(r["hi"]) << 64) | int(r["lo"])
    doc = {"name": "self", "edges": [["%032x" % u(r["from"]), "%032x" % u(r["to"]), int(r["rel_flags"])] for r in e],
           "centrality": [["%032x" % i, "%016x" % b] for i, b in as_pairs(ids, vals)]}
    p = tmp_path / "reference_self.json"
    p.write_text(json.dumps(doc))
    name, e2, want = load_case(str(p))
    assert name == "self" and np.array_equal(e2, e) and want == as_pairs(ids, vals) and load_pages(str(p)) is None
    doc["pages"] = doc["edges"][:7]
    p.write_text(json.dumps(doc))
    assert np.array_equal(load_pages(str(p)), e[:7])


@pytest.mark.skipif(not FILES, reason="no reference-produced golden vectors committed (tools/ref_golden.rs needs cargo)")
@pytest.mark.parametrize("path", FILES)
def test_oracle_matches_reference(path):
    name, e, want = load_case(path)
    ids, vals, st = hbo.faithful_run(e, load_pages(path))   # pages present: the reference's page-level tail (App. C-5)
    assert as_pairs(ids, vals) == want, name


@pytest.mark.gpu
@pytest.mark.skipif(not FILES, reason="no reference-produced golden vectors committed (tools/ref_golden.rs needs cargo)")
@pytest.mark.parametrize("path", FILES)
def test_gpu_matches_reference(path):
    from stract_amd.harmonic import EdgeListGraph, HarmonicCentrality
    name, e, want = load_case(path)
    pages = load_pages(path)
    if pages is None:
        hc = HarmonicCentrality.calculate(EdgeListGraph(e))
        assert as_pairs(*hc.arrays()) == want, name
    else:
        with _lib.Context(flags=_lib.HB_FLAG_REFERENCE_TAIL) as ctx:
            ctx.load_edges(e)
            ctx.load_tail_edges(pages)
            ctx.run()
            assert as_pairs(*ctx.results()) == want, name


@pytest.mark.skipif(not os.path.isdir(STORE), reason="no reference-written edge store committed (tools/ref_golden.rs)")
def test_column_reader_on_reference_store():
    from stract_amd import webgraph
    name, e, want = load_case(os.path.join(HERE, "golden", "reference_salted.json"))
    with webgraph.EdgeStoreReader(STORE, verify_crc=True) as r:
        got = r.read()
    # host_edges() = the stream after unique_by (store.rs:313): first occurrence of every pair, stream order kept
    seen, keep = set(), []
    for i, rec in enumerate(got):
        k = (rec["from"].tobytes(), rec["to"].tobytes())
        if k not in seen:
            seen.add(k)
            keep.append(i)
    assert np.array_equal(got[keep], e)
