"""Native webgraph column reader (include/hb_webgraph.h, stract_amd/csrc/hb_webgraph.cpp): SURVEY.md §8(f) rank 1.
CPU tests: reference-held bytes for the sstable framing and the meta.json shape, then whole fixtures written by
tests/tantivy_fixture.py (a restatement of the reference serialisers - "format unpinned", see its header)."""
import ctypes
import os
import struct
import zlib

import numpy as np
import pytest

from stract_amd import _lib, synth, webgraph
from tests import tantivy_fixture as tf


def _sst(bytes_, mode):
    lib = _lib.load()
    buf = np.frombuffer(bytes(bytes_), dtype=np.uint8)
    keys = np.zeros(4096, dtype=np.uint8)
    ranges = np.zeros(256, dtype=np.uint64)
    cnt = ctypes.c_uint64(0)
    rc = lib.hbw_debug_sstable(buf.ctypes.data, len(buf), mode, keys.ctypes.data, len(keys), ranges.ctypes.data, len(ranges),
                               ctypes.byref(cnt))
    assert rc == 0, lib.hbw_last_error(None)
    out, pos = [], 0
    for i in range(cnt.value):
        (kl,) = struct.unpack_from("<I", keys, pos)
        out.append((bytes(keys[pos + 4:pos + 4 + kl]), (int(ranges[2 * i]), int(ranges[2 * i + 1]))))
        pos += 4 + kl
    return out


def test_sstable_framing_matches_reference_bytes():
    # crates/tantivy/src/sstable/mod.rs:373-396 test_simple_sstable: the bytes the reference's writer produces
    golden = bytes([8, 0, 0, 0, 0, 16, 17, 33, 18, 19, 17, 20, 0, 0, 0, 0,
                    0, 0, 0, 0, 0, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0])
    assert [k for k, _ in _sst(golden, 0)] == [bytes([17]), bytes([17, 18, 19]), bytes([17, 20])]


def test_fixture_dictionary_round_trip():
    entries = [(b"from_host_id\0\x06", (0, 40)), (b"from_id\0\x06", (40, 77)), (b"rel_flags\0\x01", (77, 100)),
               (b"sort_score\0\x01", (100, 3000000000)), (b"x" * 40 + b"\0\x01", (3000000000, 2 ** 40))]
    assert _sst(tf.sstable_ranges(entries), 1) == entries
    # the writer restatement reproduces the reference's bytes for its own test case (keys only, no values)
    assert tf.sst_vint(300) == bytes([0xAC, 0x02]) and tf.sst_vint(127) == bytes([127])


def test_crc32_is_ieee():
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 9, 1000, 65537):
        b = rng.integers(0, 256, n, dtype=np.uint8)
        assert _lib.load().hbw_debug_crc32(b.ctypes.data if n else None, n) == (zlib.crc32(b.tobytes()) & 0xFFFFFFFF)
    # [r6] the carry-less-multiplication form (hb_webgraph.cpp crc32_clmul: 64-byte folds, 16-byte folds, Barrett reduction, the table
    # form for the last < 16 bytes) at every length around its block sizes and at odd alignments; zlib is the independent statement
    buf = rng.integers(0, 256, 5000, dtype=np.uint8)
    for n in list(range(0, 300)) + [1023, 1024, 1025, 4095, 4096, 4097]:
        for off in (0, 1, 3, 7, 13):
            b = buf[off:off + n]
            assert _lib.load().hbw_debug_crc32(b.ctypes.data if n else None, n) == (zlib.crc32(b.tobytes()) & 0xFFFFFFFF), (n, off)


def test_meta_json_reference_shape(tmp_path):
    # index_meta.rs:436-440: the JSON the reference's own test expects for an index without segments
    ref = ('{"index_settings":{"sort_by_field":{"field":"text","order":"Asc"},"docstore_compression":"lz4","docstore_blocksize":16384},'
           '"segments":[],"schema":[{"name":"text","type":"text","options":{"indexing":{"record":"position","fieldnorms":true,'
           '"tokenizer":"default"},"stored":false,"columnar":false}}],"opstamp":0}')
    (tmp_path / "meta.json").write_text(ref)
    with webgraph.EdgeStoreReader(str(tmp_path)) as r:
        assert r.num_segments() == 0 and r.total_rows() == 0
        assert len(r.read()) == 0


def test_reads_what_the_fixture_writer_wrote(tmp_path):
    g = synth.RmatGraph(10, 6000)
    e = g.edges(salt=1, salt_seed=4)
    parts = [e[:1000], e[1000:1000], e[1000:4321], e[4321:]]  # an empty segment in between
    ids = tf.write_edge_store(str(tmp_path / "edges"), parts)
    with webgraph.EdgeStoreReader(str(tmp_path / "edges"), verify_crc=True) as r:
        assert r.num_segments() == 4 and r.total_rows() == len(e)
        assert [r.segment_info(i) for i in range(4)] == [(ids[i], len(parts[i])) for i in range(4)]
        assert np.array_equal(r.read(), e)                       # stream order = segment order, then doc order
        assert np.array_equal(r.read(990, 20), e[990:1010])      # a window across a segment boundary
        assert len(r.read(len(e), 0)) == 0
        with pytest.raises(_lib.HyperballError):
            r.read(len(e) - 1, 2)


def test_corrupt_stores_are_rejected(tmp_path):
    g = synth.RmatGraph(8, 500)
    e = g.edges()
    d = str(tmp_path / "edges")
    (uid,) = tf.write_edge_store(d, [e])
    col = os.path.join(d, uid + ".col")
    good = open(col, "rb").read()

    def expect_fail(data, what, verify_crc=False):
        open(col, "wb").write(data)
        with pytest.raises(_lib.HyperballError) as ei:
            webgraph.EdgeStoreReader(d, verify_crc=verify_crc)
        assert what in str(ei.value), str(ei.value)

    expect_fail(good[:-4] + struct.pack("<I", 1336), "magic")
    flipped = bytearray(good)
    flipped[100] ^= 0x40
    expect_fail(bytes(flipped), "CRC", verify_crc=True)
    # a store whose rel_flags column uses another codec than Raw (the reference snapshot never writes one)
    body = bytearray(tf.columnar_bytes({"from_host_id": (tf.U128, e["from"]), "to_host_id": (tf.U128, e["to"]),
                                        "rel_flags": (tf.U64, e["rel_flags"])}, len(e)))
    off = 2 * (1 + 1 + 4 + 32 + 16 * len(e) + 4) + 1  # codec byte of the third column (columns sorted by name: from, rel?, to)
    names = sorted(["from_host_id", "rel_flags", "to_host_id"])
    assert names[1] == "rel_flags"
    off = (1 + 1 + 4 + 32 + 16 * len(e) + 4) + 1
    assert body[off] == 3
    body[off] = 0
    expect_fail(tf.with_footer(bytes(body)), "codec")
    # a column missing
    expect_fail(tf.with_footer(tf.columnar_bytes({"from_host_id": (tf.U128, e["from"]), "rel_flags": (tf.U64, e["rel_flags"])}, len(e))),
                "to_host_id")
    open(col, "wb").write(good)
    os.remove(os.path.join(d, "meta.json"))
    with pytest.raises(_lib.HyperballError):
        webgraph.EdgeStoreReader(d)


def test_random_dictionaries_and_segmentations(tmp_path):
    """Property-style: random column-name sets (long shared prefixes force the vint key headers, delta.rs:90-100) and
    random segment splits must read back exactly."""
    rng = np.random.default_rng(99)
    for case in range(20):
        names = set()
        while len(names) < int(rng.integers(1, 12)):
            base = "x" * int(rng.integers(0, 30)) + "".join(chr(int(c)) for c in rng.integers(97, 123, int(rng.integers(1, 25))))
            names.add(base.encode() + b"\0" + bytes([int(rng.choice([1, 6]))]))
        entries, off = [], 0
        for k in sorted(names):
            ln = int(rng.integers(0, 1 << int(rng.integers(1, 40))))
            entries.append((k, (off, off + ln)))
            off += ln
        assert _sst(tf.sstable_ranges(entries), 1) == entries
    g = synth.RmatGraph(9, 2500)
    e = g.edges(salt=1, salt_seed=8)
    for case in range(5):
        cuts = sorted(int(c) for c in rng.integers(0, len(e) + 1, int(rng.integers(0, 6))))
        parts = [e[a:b] for a, b in zip([0] + cuts, cuts + [len(e)])]
        d = str(tmp_path / ("edges%d" % case))
        tf.write_edge_store(d, parts, extra_columns=bool(case % 2))
        with webgraph.EdgeStoreReader(d, verify_crc=True) as r:
            assert r.num_segments() == len(parts)
            assert np.array_equal(r.read(), e)
            a = int(rng.integers(0, len(e)))
            b = int(rng.integers(a, len(e) + 1))
            assert np.array_equal(r.read(a, b - a), e[a:b])


def test_u128_raw_codec_datasets_of_the_reference(tmp_path):
    """crates/tantivy/src/columnar/column_values/u128_based/tests.rs: the value sets the reference runs through its `Raw` u128
    codec (the only codec its columnar writer emits for NodeID columns, column/serialize.rs:23,51) - test_serialize_and_load_simple
    (:5-14), test_empty_column_u128 (:17-32), test_small_raw_example (:123-126), get_codec_test_datasets (:150-172) and the three
    families of num_strategy (:136-142: values hugging u128::MAX, values hugging 0, anything) - as NodeID columns of an edge store,
    read back by the native reader.  The reference holds no byte vectors for this codec (its tests are round trips through its
    own reader), so this pins coverage, not bytes: every value class its tests name must survive OUR writer restatement + reader."""
    rng = np.random.default_rng(17)
    top = (1 << 128) - 1
    datasets = [[1, 2, 5], [], [9223372036854775808, 9223370937344622593], list(range(10, 10_001)), [5, 6, 7, 8, 9, 10, 99, 100],
                [5, 50, 3, 13, 1, 1000, 35], [10], [1572656989877777, 1170935903116329, 720575940379279, 0],
                [top - int(x) % 10 for x in rng.integers(0, 1 << 62, 40)], [int(x) % 10 for x in rng.integers(0, 1 << 62, 40)],
                [(int(a) << 64) | int(b) for a, b in zip(rng.integers(0, 1 << 63, 5000, dtype=np.uint64) * 2 + 1, rng.integers(0, 1 << 63, 5000, dtype=np.uint64))]]
    parts = []
    for vals in datasets:
        e = np.zeros(len(vals), dtype=_lib.EDGE)
        for i, v in enumerate(vals):
            w = vals[len(vals) - 1 - i]
            e["from"]["lo"][i], e["from"]["hi"][i] = v & 0xFFFFFFFFFFFFFFFF, v >> 64
            e["to"]["lo"][i], e["to"]["hi"][i] = w & 0xFFFFFFFFFFFFFFFF, w >> 64
            e["rel_flags"][i] = (v ^ (v >> 64)) & 0xFFFFFFFFFFFFFFFF
        parts.append(e)
    tf.write_edge_store(str(tmp_path / "edges"), parts)
    with webgraph.EdgeStoreReader(str(tmp_path / "edges"), verify_crc=True) as r:
        assert r.num_segments() == len(parts) and r.total_rows() == sum(len(p_) for p_ in parts)
        got = r.read()
    at = 0
    for e in parts:
        assert np.array_equal(got[at:at + len(e)], e)
        at += len(e)


def test_page_level_id_columns(tmp_path):
    """HBW_PAGE_IDS: the documents' page-level `from_id` / `to_id` (webgraph/schema.rs:132-180) next to the host-level
    ids - what the reference's tail mode queries (harmonic.rs:82-87)."""
    g = synth.RmatGraph(10, 4_000)
    host = g.edges(salt=1, salt_seed=2)
    page = host.copy()
    page["from"]["hi"] ^= np.uint64(0x55)
    page["to"]["lo"] += np.uint64(3)
    cut = [0, 1000, len(host)]
    tf.write_edge_store(str(tmp_path / "e"), [host[a:b] for a, b in zip(cut, cut[1:])], page_segments=[page[a:b] for a, b in zip(cut, cut[1:])])
    with webgraph.EdgeStoreReader(str(tmp_path / "e"), verify_crc=True, page_ids=True) as r:
        assert np.array_equal(r.read(), host)
        assert np.array_equal(r.read(page_level=True), page)
        assert np.array_equal(r.read(990, 20, page_level=True), page[990:1010])   # across the segment boundary
    with webgraph.EdgeStoreReader(str(tmp_path / "e")) as r:   # not opened: the page-level read refuses
        with pytest.raises(_lib.HyperballError):
            r.read(page_level=True)
    tf.write_edge_store(str(tmp_path / "bare"), [host], extra_columns=False)
    with pytest.raises(_lib.HyperballError):
        webgraph.EdgeStoreReader(str(tmp_path / "bare"), page_ids=True)       # the store has no such columns
    with webgraph.EdgeStoreReader(str(tmp_path / "bare")) as r:
        assert np.array_equal(r.read(), host)


def test_directory_footer_cases_of_the_reference(tmp_path):
    """crates/tantivy/src/directory/footer.rs:169-235 - the reference's own footer tests, replayed on the reader: a file
    that ends in [footer_len u32][magic u32] with a wrong magic, a footer longer than the file, or longer than
    FOOTER_MAX_LEN = 50 000 must be refused."""
    g = synth.RmatGraph(6, 100)
    d = str(tmp_path / "edges")
    (uid,) = tf.write_edge_store(d, [g.edges()])
    col = os.path.join(d, uid + ".col")
    for data, what in ((struct.pack("<II", 0, 5555), "magic"),                      # test_deserialize_footer_missing_magic_byte
                       (struct.pack("<II", 100, 1337), "footer length"),            # test_deserialize_footer_wrong_filesize
                       (b"\0" * 60000 + struct.pack("<II", 50001, 1337), "footer length")):   # test_deserialize_too_large_footer
        open(col, "wb").write(data)
        with pytest.raises(_lib.HyperballError) as ei:
            webgraph.EdgeStoreReader(d)
        assert what in str(ei.value), str(ei.value)


def test_vint_lengths_of_the_reference_and_long_keys():
    """crates/tantivy/src/sstable/vint.rs:47-60 (test_vint): encoded lengths; then keys longer than 15 bytes, whose
    keep/add lengths are written as 0x01 + two vints (sstable/mod.rs:293-316), through the native dictionary reader."""
    for val, length in ((0, 1), (17, 1), (127, 1), (128, 2), (123423418, 4)):
        assert len(tf.sst_vint(val)) == length
    for i in range(1, 63):
        assert len(tf.sst_vint(1 << i)) == i // 7 + 1 and len(tf.sst_vint((1 << i) + 1)) == i // 7 + 1
    keys = [b"k" * 40 + bytes([c]) + b"\0\x06" for c in range(1, 30)]        # add >= 16 on the first, keep >= 16 after
    keys += [b"z" * 200 + b"tail" + bytes([c]) for c in range(1, 5)]          # two-byte vints
    entries, at = [], 0
    for i, k in enumerate(keys):                                              # RangeSSTable: ranges are contiguous
        entries.append((k, (at, at + 7 + 300 * i)))
        at += 7 + 300 * i
    got = _sst(tf.sstable_ranges(entries), 1)
    assert got == entries


def test_zstd_compressed_dictionary_block():
    """sstable/delta.rs:55-72: a block above 2048 bytes is written zstd-compressed (level 3) when that is smaller.  The
    native reader decompresses it through libzstd.so.1 (dlopen)."""
    if tf.zstd_compress(b"x" * 100) is None:
        pytest.skip("libzstd.so.1 not loadable here")
    keys = [b"%05d_column_with_a_rather_long_and_repetitive_name" % i + b"\0\x06" for i in range(70)]   # little shared prefix: ~3.3 KB
    entries, at = [], 0
    for i, k in enumerate(keys):
        entries.append((k, (at, at + 16 + i)))
        at += 16 + i
    raw_len = sum(len(k) for k in keys)
    sst = tf.sstable_ranges(entries)
    assert sst[4] == 1 and len(sst) < raw_len          # the block is flagged compressed and is smaller than its keys alone
    lib = _lib.load()
    buf = np.frombuffer(bytes(sst), dtype=np.uint8)
    keys_out = np.zeros(16384, dtype=np.uint8)
    ranges = np.zeros(512, dtype=np.uint64)
    cnt = ctypes.c_uint64(0)
    rc = lib.hbw_debug_sstable(buf.ctypes.data, len(buf), 1, keys_out.ctypes.data, len(keys_out), ranges.ctypes.data, len(ranges), ctypes.byref(cnt))
    assert rc == 0, lib.hbw_last_error(None)
    got, pos = [], 0
    for i in range(cnt.value):
        (kl,) = struct.unpack_from("<I", keys_out, pos)
        got.append((bytes(keys_out[pos + 4:pos + 4 + kl]), (int(ranges[2 * i]), int(ranges[2 * i + 1]))))
        pos += 4 + kl
    assert got == entries
    # a corrupted compressed block is refused, not mis-read
    bad = bytearray(sst)
    bad[20] ^= 0xFF
    buf2 = np.frombuffer(bytes(bad), dtype=np.uint8)
    assert lib.hbw_debug_sstable(buf2.ctypes.data, len(buf2), 1, keys_out.ctypes.data, len(keys_out), ranges.ctypes.data, len(ranges), ctypes.byref(cnt)) != 0


def test_crc32_pieces_and_combine_equal_zlib():
    """the CRC of a large .col body is computed in 64 MiB pieces on all cores and combined (crc of a concatenation from the
    crcs of its parts); zlib.crc32 is the independent statement of the same polynomial"""
    import zlib
    rng = np.random.default_rng(5)
    for nbytes in (0, 1, 7, 1 << 20, (130 << 20) + 12345, (64 << 20) * 3):
        buf = rng.integers(0, 256, nbytes, dtype=np.uint8)
        got = _lib.load().hbw_debug_crc32(buf.ctypes.data_as(ctypes.c_void_p), nbytes)
        assert got == zlib.crc32(buf.tobytes()), nbytes


def test_streamed_fixture_writer_equals_the_plain_one(tmp_path):
    """write_edge_store_streamed (one assembled buffer per segment, generator input, pluggable CRC) must write the very
    files write_edge_store(extra_columns=False) writes"""
    g = synth.RmatGraph(10, 5000)
    e = g.edges(salt=1, salt_seed=3)
    segs = [e[:1000], e[1000:1001], e[1001:]]
    a, b = tmp_path / "plain", tmp_path / "streamed"
    ids_a = tf.write_edge_store(str(a), segs, extra_columns=False)
    ids_b = tf.write_edge_store_streamed(str(b), (s for s in segs),
                                         crc32=lambda buf: _lib.load().hbw_debug_crc32(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))
    assert ids_a == ids_b
    for name in sorted(os.listdir(a)):
        assert open(a / name, "rb").read() == open(b / name, "rb").read(), name
    # several segments in the making at once, produced on demand (what bench.py's end-to-end leg does): the same files again
    c = tmp_path / "threaded"
    made = []
    ids_c = tf.write_edge_store_streamed(str(c), ((lambda s=s: (made.append(len(s)), s)[1]) for s in segs), workers=3)
    assert ids_c == ids_a and sorted(made) == sorted(len(s) for s in segs)
    for name in sorted(os.listdir(a)):
        assert open(a / name, "rb").read() == open(c / name, "rb").read(), name
