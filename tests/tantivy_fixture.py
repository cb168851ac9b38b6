"""Writes a Stract webgraph edge store (`<webgraph>/edges`: meta.json + one `.col` file per segment) for tests of the
native reader (include/hb_webgraph.h).  TEST INFRASTRUCTURE: a plain-Python restatement of the reference's
serialisers, each function citing what it follows (paths under /root/reference/crates/tantivy/src):

  columnar file   columnar/columnar/writer/mod.rs:264-300 (columns sorted by (name, type)), writer/serializer.rs:20-69
  one column      columnar/column/serialize.rs:17-26,47-56 + column_index/serialize.rs:21-36 (cardinality Full = 0)
  u128 values     column_values/u128_based/mod.rs:97-102 (codec byte 0 = Raw) + raw.rs:94-113
  u64 values      column_values/u64_based/mod.rs:27-32,121-126 (codec byte 3 = Raw) + raw.rs
  dictionary      sstable/mod.rs:231-316 (Writer), delta.rs:45-110 (blocks, keep/add), value/range.rs:44-63, vint.rs:3-16
  file footer     directory/footer.rs:33-41 (JSON + len + 1337), CRC-32 of the body (footer.rs FooterProxy)
  meta.json       index/index_meta.rs:215-225,325-342,436-440

"Format unpinned": nothing here was checked against a file written by the reference (no Rust toolchain in this
image); tools/ref_golden.rs writes a whole store (`store_fixture`) with the reference itself."""
import json
import os
import struct
import uuid
import zlib

import numpy as np

U64, U128 = 1, 6  # ColumnType codes, columnar/columnar/column_type.rs:13-21


def sst_vint(v):
    """sstable/vint.rs:3-16: 7 bits per byte, continue bit on all but the last byte."""
    out = bytearray()
    while True:
        b = v & 127
        v >>= 7
        if v == 0:
            out.append(b)
            return bytes(out)
        out.append(b | 128)


def zstd_compress(data, level=3):
    """ZSTD_compress through libzstd.so.1 (what `zstd::bulk::Compressor::new(3)` calls); None if the library is absent."""
    import ctypes
    try:
        z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(data))
    dst = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(dst, cap, data, len(data), level)
    if z.ZSTD_isError(n):
        return None
    return dst.raw[:n]


def sstable_ranges(entries, block_len=4000):
    """Dictionary<RangeSSTable> bytes for sorted [(key, (start, end))].  Follows Writer::insert / DeltaWriter::flush_block /
    Writer::finish; blocks above 2048 bytes are zstd-compressed like the reference does (delta.rs:55-72) when libzstd.so.1
    can be loaded."""
    out = bytearray()
    block, vals, prev_key, nblocks = bytearray(), [], b"", 0

    def flush():
        nonlocal block, vals, nblocks
        if not block:
            return
        vb = bytearray(sst_vint(len(vals)))
        prev = 0
        for v in vals:
            vb += sst_vint(v - prev)
            prev = v
        total = len(vb) + len(block)
        packed = zstd_compress(bytes(vb) + bytes(block)) if total > 2048 else None   # delta.rs:55-72: level 3, kept if smaller
        if packed is not None and len(packed) < total:
            out.extend(struct.pack("<I", len(packed) + 1))
            out.append(1)
            out.extend(packed)
        else:
            assert total <= 2048 or packed is not None, "the reference would zstd-compress this block (delta.rs:58); libzstd not loadable"
            out.extend(struct.pack("<I", total + 1))
            out.append(0)
            out.extend(vb)
            out.extend(block)
        block, vals = bytearray(), []
        nblocks += 1

    for key, (start, end) in entries:
        keep = 0
        while keep < min(len(prev_key), len(key)) and prev_key[keep] == key[keep]:
            keep += 1
        add = len(key) - keep
        assert not prev_key or key > prev_key, "keys must increase"
        if keep < 16 and add < 16:
            block.append(keep | (add << 4))
        else:
            block.append(1)
            block.extend(sst_vint(keep))
            block.extend(sst_vint(add))
        block.extend(key[keep:])
        if vals:
            assert vals[-1] == start
            vals.append(end)
        else:
            vals.extend([start, end])
        prev_key = key
        if len(block) > block_len:
            flush()
            prev_key = b""
    flush()
    out.extend(struct.pack("<I", 0))  # end marker
    offset = len(out)
    assert nblocks <= 1, "multi-block dictionaries need the fst index (sstable_index_v3.rs:281-303)"
    out.extend(struct.pack("<Q", 0))           # fst length: 0 = no index (<= 1 block)
    out.extend(struct.pack("<Q", offset))      # index start offset
    out.extend(struct.pack("<Q", len(entries)))
    out.extend(struct.pack("<I", 3))           # SSTABLE_VERSION
    return bytes(out)


def column_bytes(ctype, values):
    """[column index][column values][column_index_num_bytes u32]"""
    n = len(values)
    out = bytearray([0])  # Cardinality::Full
    if ctype == U128:
        v = np.ascontiguousarray(values)  # structured (lo, hi) little-endian = u128 little-endian
        out.append(0)  # u128 CodecType::Raw
        out += struct.pack("<I", n)
        for pick in (np.min, np.max):  # min / max of the u128 values: extreme high half, then the extreme low half among those
            hi = pick(v["hi"]) if n else 0
            lo = pick(v["lo"][v["hi"] == hi]) if n else 0
            out += struct.pack("<QQ", int(lo), int(hi))
        out += v.tobytes()
    else:
        v = np.ascontiguousarray(values, dtype="<u8")
        out.append(3)  # u64 CodecType::Raw
        out += struct.pack("<I", n)
        out += struct.pack("<QQ", int(v.min()) if n else 0, int(v.max()) if n else 0)
        out += v.tobytes()
    out += struct.pack("<I", 1)  # the column index is the single cardinality byte
    return bytes(out)


def columnar_bytes(columns, num_rows):
    """columns: {name: (type code, values)}"""
    data, entries = bytearray(), []
    for name, (ctype, values) in sorted(columns.items(), key=lambda kv: (kv[0].encode(), kv[1][0])):
        start = len(data)
        data += column_bytes(ctype, values)
        entries.append((name.encode().replace(b"\0", b"0") + b"\0" + bytes([ctype]), (start, len(data))))
    sst = sstable_ranges(entries)
    return bytes(data) + sst + struct.pack("<QI", len(sst), num_rows) + struct.pack("<I", 1) + bytes([2, 113, 119, 66])


def with_footer(body):
    js = json.dumps({"version": {"major": 0, "minor": 23, "patch": 0, "index_format_version": 6},
                     "crc": zlib.crc32(body) & 0xFFFFFFFF}, separators=(",", ":")).encode()
    return body + js + struct.pack("<II", len(js), 1337)


def _column_parts(ctype, values):
    """column_bytes as a list of buffers (header bytes, the value array itself, trailer) - no copy of the values"""
    b = column_bytes(ctype, values[:0])  # header layout from the one implementation above, values spliced in below
    n = len(values)
    if ctype == U128:
        v = np.ascontiguousarray(values)
        head = bytearray([0, 0]) + struct.pack("<I", n)
        for pick in (np.min, np.max):
            hi = pick(v["hi"]) if n else 0
            lo = pick(v["lo"][v["hi"] == hi]) if n else 0
            head += struct.pack("<QQ", int(lo), int(hi))
    else:
        v = np.ascontiguousarray(values, dtype="<u8")
        head = bytearray([0, 3]) + struct.pack("<I", n) + struct.pack("<QQ", int(v.min()) if n else 0, int(v.max()) if n else 0)
    assert len(head) == len(b) - 4
    return [bytes(head), v.view(np.uint8).reshape(-1), struct.pack("<I", 1)]


def _write_streamed_segment(path, i, edges, crc32):
    """One segment file of write_edge_store_streamed; returns (meta entry, hex id)."""
    if callable(edges):
        edges = edges()
    sid = uuid.UUID(int=(0xA5C4DFCBDFE645089129E308E26D5500 + i))
    n = len(edges)
    cols = {"from_host_id": (U128, edges["from"]), "to_host_id": (U128, edges["to"]), "rel_flags": (U64, edges["rel_flags"])}
    parts, entries, at = [], [], 0
    for name, (ctype, values) in sorted(cols.items(), key=lambda kv: (kv[0].encode(), kv[1][0])):
        start = at
        for piece in _column_parts(ctype, values):
            parts.append(piece)
            at += len(piece)
        entries.append((name.encode() + b"\0" + bytes([ctype]), (start, at)))
    sst = sstable_ranges(entries)
    tail = sst + struct.pack("<QI", len(sst), n) + struct.pack("<I", 1) + bytes([2, 113, 119, 66])
    body = np.empty(at + len(tail), dtype=np.uint8)
    pos = 0
    for piece in parts + [tail]:
        k = len(piece)
        body[pos:pos + k] = np.frombuffer(piece, dtype=np.uint8) if isinstance(piece, (bytes, bytearray)) else piece
        pos += k
    crc = (crc32(body) if crc32 else zlib.crc32(body)) & 0xFFFFFFFF
    js = json.dumps({"version": {"major": 0, "minor": 23, "patch": 0, "index_format_version": 6}, "crc": crc}, separators=(",", ":")).encode()
    with open(os.path.join(path, sid.hex + ".col"), "wb") as f:
        f.write(body)
        f.write(js + struct.pack("<II", len(js), 1337))
    return {"segment_id": str(sid), "max_doc": n, "deletes": None}, sid.hex


def write_edge_store_streamed(path, segments, crc32=None, workers=1):
    """The same store as write_edge_store(extra_columns=False), for BASELINE-size streams: `segments` may be a generator (one
    EDGE array at a time, never the whole stream) of arrays or of zero-argument callables that produce them (so that `workers`
    threads can generate, assemble, checksum and write different segments at the same time - the heavy steps are numpy copies,
    C calls and file writes, which all release the interpreter lock; at most `workers` segments are alive at once); every
    segment body is assembled once in one buffer, and `crc32` may be a faster CRC-32 (IEEE) of a uint8 array than zlib's single
    thread (the library's hbw_debug_crc32: pieces on all cores).  The files do not depend on `workers`."""
    os.makedirs(path, exist_ok=True)
    if workers <= 1:
        done = [_write_streamed_segment(path, i, edges, crc32) for i, edges in enumerate(segments)]
    else:
        import threading
        from concurrent.futures import ThreadPoolExecutor
        slots = threading.Semaphore(workers)

        def job(i, edges):
            try:
                return _write_streamed_segment(path, i, edges, crc32)
            finally:
                slots.release()

        futures = []
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for i, edges in enumerate(segments):
                slots.acquire()  # (a generator of arrays is only advanced when a worker is free)
                futures.append(pool.submit(job, i, edges))
        done = [f.result() for f in futures]
    metas = [m for m, _ in done]
    ids = [h for _, h in done]
    meta = {"index_settings": {"sort_by_field": {"field": "sort_score", "order": "Asc"}, "docstore_compression": "lz4",
                               "docstore_blocksize": 16384},
            "segments": metas, "schema": [{"name": "from_host_id", "type": "u128", "options": {"columnar": True}}], "opstamp": 7}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    return ids


def write_edge_store(path, segments, extra_columns=True, page_segments=None):
    """segments: list of EDGE record arrays (stract_amd._lib.EDGE), one tantivy segment each.  Returns the uuids.
    page_segments: per segment, EDGE arrays whose from / to are the documents' page-level `from_id` / `to_id`
    (webgraph/schema.rs:132-180); default: decoy values, so that a reader picking the wrong column is caught."""
    os.makedirs(path, exist_ok=True)
    metas, ids = [], []
    for i, edges in enumerate(segments):
        sid = uuid.UUID(int=(0xA5C4DFCBDFE645089129E308E26D5500 + i))
        n = len(edges)
        cols = {"from_host_id": (U128, edges["from"]), "to_host_id": (U128, edges["to"]), "rel_flags": (U64, edges["rel_flags"])}
        if extra_columns:  # other columnar fields of the webgraph schema (webgraph/schema.rs), ignored by the reader
            cols["from_id"] = (U128, page_segments[i]["from"] if page_segments else edges["to"])
            cols["to_id"] = (U128, page_segments[i]["to"] if page_segments else edges["from"])
            cols["sort_score"] = (U64, np.arange(n, dtype=np.uint64))
        with open(os.path.join(path, sid.hex + ".col"), "wb") as f:
            f.write(with_footer(columnar_bytes(cols, n)))
        metas.append({"segment_id": str(sid), "max_doc": n, "deletes": None})
        ids.append(sid.hex)
    meta = {"index_settings": {"sort_by_field": {"field": "sort_score", "order": "Asc"}, "docstore_compression": "lz4",
                               "docstore_blocksize": 16384},
            "segments": metas, "schema": [{"name": "from_host_id", "type": "u128", "options": {"columnar": True}}], "opstamp": 7}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f, separators=(",", ":"))
    return ids
