"""The native speedy_kv store writer (include/hb_store.h) against an independent Python reader (tests/speedy_kv_reader.py).
CPU only: the writer is host code.  Format unpinned (no reference-written store exists in this image): these tests prove that
writer and reader, written separately from the same format descriptions, agree - and pin the parts the reference itself
defines (file set, BlobPointer records, key order, bloom sizing)."""
import json
import os
import struct

import numpy as np
import pytest

from stract_amd import _lib
from tests import speedy_kv_reader as kv


def _ids(rng, count):
    """NodeIDs of every bincode length class: 1, 3, 5, 9 and 17 bytes"""
    pool = set()
    for bits, share in ((7, 0.02), (16, 0.08), (32, 0.15), (64, 0.25), (128, 0.5)):
        want = max(2, int(count * share))
        while want:
            v = int.from_bytes(rng.bytes(16), "little") >> (128 - bits)
            if v not in pool:
                pool.add(v)
                want -= 1
    pool.update((0, 250, 251, 65535, 65536, (1 << 32) - 1, 1 << 32, (1 << 64) - 1, 1 << 64, (1 << 128) - 1))
    ints = list(pool)
    rng.shuffle(ints)
    return ints


@pytest.mark.parametrize("count", [40, 5000])
def test_store_round_trip(tmp_path, count):
    rng = np.random.default_rng(count)
    ints = _ids(rng, count)
    ids = kv.ints_to_ids(ints, _lib.U128)
    vals = rng.random(len(ints)) * 3.0
    vals[:3] = (0.0, 5e-324, 1.7976931348623157e308)
    ranks = rng.permutation(len(ints)).astype(np.uint64)
    ranks[0] = np.uint64((1 << 64) - 1)
    _lib.store_harmonic(str(tmp_path), ids, vals, ranks)
    assert sorted(os.listdir(tmp_path)) == ["harmonic", "harmonic_rank"]
    for name, kind, values in (("harmonic", "f64", vals), ("harmonic_rank", "u64", ranks)):
        folder = os.path.join(str(tmp_path), name)
        files = sorted(os.listdir(folder))
        meta = json.load(open(os.path.join(folder, "meta.json")))
        (uuid,) = meta["segments"]
        assert files == sorted([uuid + e for e in (".bid", ".blm", ".blobs", ".ids")] + ["meta.json"])
        assert len(uuid) == 36 and uuid[14] == "4" and uuid[19] in "89ab"
        db = kv.Db(folder, kind, str(tmp_path))
        assert len(db) == len(ints)
        want = {i: v for i, v in zip(ints, values.tolist())}
        got = list(db.items())
        # entries lie in ascending order of the key BYTES (BTreeMap<Vec<u8>, _>), not of the ids
        keys = [kv.varint_encode(i) for i, _ in got]
        assert keys == sorted(keys) and len(set(keys)) == len(keys)
        for i, v in got:
            w = want[i]
            assert (struct.pack("<d", v) == struct.pack("<d", w)) if kind == "f64" else (v == w)
        assert len(got) == len(want)
        for i in ints[:400]:
            v = db.get(i)
            assert v is not None and ((struct.pack("<d", v) == struct.pack("<d", want[i])) if kind == "f64" else v == want[i])
        missing = [int.from_bytes(rng.bytes(16), "little") for _ in range(300)] + [1, 252, 65537]
        assert all(db.get(i) is None for i in missing if i not in want)
        seg = db.segments[0]
        # the fst answers what the bloom filter lets through, and holds exactly the keys: value = BlobId = position
        assert [k for k, _ in seg.fst.items()] == keys
        assert [b for _, b in seg.fst.items()] == list(range(len(keys)))
        assert all(seg.bloom.contains(k) for k in keys)
        # bloom sizing: bloom/src/lib.rs:38-48 with fp = 0.01
        import math
        bits = math.ceil(len(ints) * math.log(0.01) / (-8.0 * math.log(2.0) ** 2))
        assert seg.bloom.bits == bits and seg.bloom.num_hashes == max(math.ceil(bits / len(ints) * math.log(2.0)), 1)
        # BlobPointer records tile the blob file
        at = 0
        for b in range(len(keys)):
            ks, ke, vs, ve = seg.pointer(b)
            assert ks == at and ke == vs and ve >= vs
            at = ve
        assert at == len(seg.blobs)


def test_store_many_keys_index_nodes_and_wide_deltas(tmp_path):
    """200 k random 128-bit ids: the second trie level has 256 transitions (index table, count byte 1 = 256), addresses need
    3-byte deltas, BlobIds need 3-byte outputs"""
    rng = np.random.default_rng(7)
    n = 200_000
    ids = np.zeros(n, dtype=_lib.U128)
    ids["lo"] = rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64)
    ids["hi"] = rng.integers(1, 1 << 63, n, dtype=np.uint64)
    vals = rng.random(n)
    _lib.store_write(str(tmp_path / "db"), ids, vals)
    db = kv.Db(str(tmp_path / "db"), "f64", str(tmp_path))
    assert len(db) == n
    seg = db.segments[0]
    _, _, root = seg.fst.node(seg.fst.root)
    assert [t[0] for t in root] == [254]
    _, _, second = seg.fst.node(root[0][2])
    assert len(second) == 256
    ints = kv.ids_to_ints(ids)
    pick = rng.integers(0, n, 2000)
    for j in pick.tolist():
        assert db.get(ints[j]) == vals[j]
    total = 0
    prev = b""
    for k, b in seg.fst.items():
        assert k > prev and b == total
        prev = k
        total += 1
    assert total == n


def test_store_empty_duplicate_and_errors(tmp_path):
    ids = np.zeros(0, dtype=_lib.U128)
    _lib.store_write(str(tmp_path / "empty"), ids, np.zeros(0))
    assert json.load(open(tmp_path / "empty" / "meta.json")) == {"segments": []}
    assert os.listdir(tmp_path / "empty") == ["meta.json"]
    db = kv.Db(str(tmp_path / "empty"), "f64", str(tmp_path))
    assert len(db) == 0 and db.get(5) is None
    dup = kv.ints_to_ids([5, 9, 5], _lib.U128)
    with pytest.raises(_lib.HyperballError) as e:
        _lib.store_write(str(tmp_path / "dup"), dup, np.zeros(3))
    assert e.value.code == _lib.HB_ERR_INVALID and "duplicate" in str(e.value)
    with pytest.raises(TypeError):
        _lib.store_write(str(tmp_path / "x"), dup[:2], np.zeros(2, dtype=np.int32))
    blocker = tmp_path / "file"
    blocker.write_text("x")
    with pytest.raises(_lib.HyperballError) as e:
        _lib.store_write(str(blocker / "sub"), dup[:2], np.zeros(2))
    assert e.value.code == _lib.HB_ERR_IO
    # one entry: bloom of 2 bits / 2 hashes (ceil(1 * ln 0.01 / (-8 ln^2 2)) = 2)
    one = kv.ints_to_ids([1 << 100], _lib.U128)
    _lib.store_write(str(tmp_path / "one"), one, np.array([7], dtype=np.uint64))
    db = kv.Db(str(tmp_path / "one"), "u64", str(tmp_path))
    assert db.get(1 << 100) == 7 and db.segments[0].bloom.bits == 2 and db.segments[0].bloom.num_hashes == 2


def test_bincode_integer_classes():
    for v, n in ((0, 1), (250, 1), (251, 3), (65535, 3), (65536, 5), ((1 << 32) - 1, 5), (1 << 32, 9), ((1 << 64) - 1, 9), (1 << 64, 17)):
        enc = kv.varint_encode(v)
        assert len(enc) == n and kv.varint_decode(enc) == (v, n)


def _segment_files(folder):
    (uuid,) = json.load(open(os.path.join(folder, "meta.json")))["segments"]
    return {ext: open(os.path.join(folder, uuid + ext), "rb").read() for ext in (".ids", ".blm", ".bid", ".blobs")}


@pytest.mark.parametrize("kind", ["hashes", "mixed_lengths", "consecutive", "two_groups"])
def test_parallel_fst_equals_one_builder(tmp_path, kind):
    """The .ids file is built as independent sub-tries (one per 3-byte key prefix, in parallel) hung below a sequentially
    written top; fst stores node addresses as distances, so the bytes must equal what ONE builder fed with all keys in order
    writes (HB_STORE_FST=sequential).  Every file of the segment is compared byte for byte."""
    rng = np.random.default_rng(len(kind))
    if kind == "hashes":        # 120 k random 128-bit ids: ~55 k groups, 256-way top nodes
        ints = [int.from_bytes(rng.bytes(16), "little") | (1 << 127) for _ in range(120_000)]
    elif kind == "mixed_lengths":
        ints = _ids(rng, 5_000)   # (the 1-byte class has only 251 values: the helper cannot draw more than that share)
    elif kind == "consecutive":  # long shared prefixes inside one group, keys that are neighbours in every byte
        base = (0xABCDEF << 100) | (7 << 64)
        ints = [base + i for i in range(70_000)] + [base + (1 << 40) + 3 * i for i in range(5_000)]
    else:                        # exactly two groups of one key each + short keys in front
        ints = [0, 17, 300, 70_000, (1 << 127) + 5, (1 << 127) + (1 << 20)]
    ints = list(dict.fromkeys(ints))
    ids = kv.ints_to_ids(ints, _lib.U128)
    ranks = rng.permutation(len(ints)).astype(np.uint64)
    _lib.store_write(str(tmp_path / "par"), ids, ranks)
    os.environ["HB_STORE_FST"] = "sequential"
    try:
        _lib.store_write(str(tmp_path / "seq"), ids, ranks)
    finally:
        del os.environ["HB_STORE_FST"]
    a, b = _segment_files(str(tmp_path / "par")), _segment_files(str(tmp_path / "seq"))
    for ext in a:
        assert a[ext] == b[ext], (kind, ext, len(a[ext]), len(b[ext]))
    db = kv.Db(str(tmp_path / "par"), "u64", str(tmp_path))
    assert len(db) == len(ints)
    for j in rng.integers(0, len(ints), 300).tolist():
        assert db.get(ints[j]) == int(ranks[j])


def test_store_harmonic_shares_the_key_files(tmp_path):
    """both databases hold the same keys in the same order: .ids and .blm are built once and must be identical files"""
    rng = np.random.default_rng(99)
    ints = _ids(rng, 3000)
    ids = kv.ints_to_ids(ints, _lib.U128)
    _lib.store_harmonic(str(tmp_path), ids, rng.random(len(ints)), rng.permutation(len(ints)).astype(np.uint64))
    a, b = _segment_files(str(tmp_path / "harmonic")), _segment_files(str(tmp_path / "harmonic_rank"))
    assert a[".ids"] == b[".ids"] and a[".blm"] == b[".blm"] and a[".blobs"] != b[".blobs"]


def test_store_never_reports_ok_without_a_meta(tmp_path):
    """meta.json that cannot be written must fail the call (it used to return HB_OK and leave segment files behind);
    an existing database with segments is refused instead of being orphaned"""
    ids = kv.ints_to_ids([3, 1 << 90], _lib.U128)
    vals = np.array([0.5, 0.25])
    for count in (2, 0):
        d = tmp_path / ("blocked%d" % count)
        (d / "meta.json").mkdir(parents=True)       # a DIRECTORY where the file must go: rename() over it fails
        with pytest.raises(_lib.HyperballError) as e:
            _lib.store_write(str(d), ids[:count], vals[:count])
        assert e.value.code == _lib.HB_ERR_IO and "meta.json" in str(e.value)
        assert not os.path.exists(d / "meta.json.tmp")
    good = tmp_path / "good"
    _lib.store_write(str(good), ids, vals)
    before = sorted(os.listdir(good))
    with pytest.raises(_lib.HyperballError) as e:
        _lib.store_write(str(good), ids, vals)
    assert e.value.code == _lib.HB_ERR_INVALID and "already lists segments" in str(e.value)
    assert sorted(os.listdir(good)) == before       # nothing added, nothing orphaned
    empty = tmp_path / "was_empty"
    _lib.store_write(str(empty), ids[:0], vals[:0])  # {"segments": []} may be filled later
    _lib.store_write(str(empty), ids, vals)
    assert kv.Db(str(empty), "f64", str(tmp_path)).get(3) == 0.5


def test_failed_store_harmonic_leaves_nothing_and_can_be_retried(tmp_path):
    """a call that fails removes the segment files it wrote, and no meta.json appears before the files of BOTH databases
    are complete - a failure while `harmonic_rank` is written must not leave a finished `harmonic` behind that makes the retry into
    the same output directory fail with "already lists segments" (Db::open_or_create would simply have gone on)."""
    ids = kv.ints_to_ids([5, 9, 1 << 100, (1 << 127) + 3], _lib.U128)
    vals = np.array([0.5, 0.25, 0.125, 1.0])
    ranks = np.array([1, 2, 3, 0], dtype=np.uint64)
    out = tmp_path / "centrality"
    (out / "harmonic_rank" / "meta.json").mkdir(parents=True)  # the SECOND database cannot get its meta.json
    with pytest.raises(_lib.HyperballError) as e:
        _lib.store_harmonic(str(out), ids, vals, ranks)
    assert e.value.code == _lib.HB_ERR_IO
    assert os.listdir(out / "harmonic") == []                                   # no segment files, no meta.json
    assert os.listdir(out / "harmonic_rank") == ["meta.json"]                   # only the obstacle itself
    os.rmdir(out / "harmonic_rank" / "meta.json")
    _lib.store_harmonic(str(out), ids, vals, ranks)                             # the retry into the same directory succeeds
    assert kv.Db(str(out / "harmonic"), "f64", str(tmp_path)).get(9) == 0.25
    assert kv.Db(str(out / "harmonic_rank"), "u64", str(tmp_path)).get(1 << 100) == 3


def test_published_vectors_of_the_restated_formats(tmp_path):
    """Pins that do not come from this repository's own reading of the formats:
      * bincode's documented variable-length integer encoding (bincode docs, `config::standard()` / VarintEncoding: u < 251 one
        byte; 251 + u16; 252 + u32; 253 + u64; 254 + u128, little endian) at every class boundary - the key bytes in .blobs;
      * XXH3: the published hashes of the empty input (xxHash's own sanity vectors: XXH3_64bits = 2D06800538D394C2,
        XXH3_128bits = 99AA06D3014798D8 6001C324468D497F), so the vendored header is the real algorithm; and the documented
        property that a secret generated from a seed reproduces the seeded variant for inputs above 240 bytes
        (xxhash.h, XXH3_generateSecret_fromSeed) - the route by which `const_custom_default_secret(42)` is restated."""
    import ctypes
    import subprocess
    want = {0: b"\x00", 250: b"\xfa", 251: b"\xfb\xfb\x00", 65535: b"\xfb\xff\xff", 65536: b"\xfc\x00\x00\x01\x00",
            (1 << 32) - 1: b"\xfc\xff\xff\xff\xff", 1 << 32: b"\xfd\x00\x00\x00\x00\x01\x00\x00\x00",
            (1 << 64) - 1: b"\xfd" + b"\xff" * 8, 1 << 64: b"\xfe" + b"\x00" * 8 + b"\x01" + b"\x00" * 7,
            (1 << 128) - 1: b"\xfe" + b"\xff" * 16}
    ints = list(want)
    ids = kv.ints_to_ids(ints, _lib.U128)
    _lib.store_write(str(tmp_path / "db"), ids, np.arange(len(ints), dtype=np.float64))
    f = _segment_files(str(tmp_path / "db"))
    keys = []
    for i in range(len(ints)):
        ks, ke, vs, ve = struct.unpack_from("<QQQQ", f[".bid"], 32 * i)
        keys.append(f[".blobs"][ks:ke])
        assert ve - vs == 8 and vs == ke
    assert sorted(keys) == keys and set(keys) == set(want.values())
    for i, k in want.items():
        assert kv.varint_encode(i) == k
    src = tmp_path / "x.c"
    src.write_text('#define XXH_INLINE_ALL\n#include "%s"\n#include <string.h>\n'
                   "unsigned long long h64(const void *p, unsigned long n) { return XXH3_64bits(p, n); }\n"
                   "void h128(const void *p, unsigned long n, unsigned long long *o) { XXH128_hash_t h = XXH3_128bits(p, n); o[0] = h.high64; o[1] = h.low64; }\n"
                   "int secret_matches_seed(const void *p, unsigned long n, unsigned long long seed) {\n"
                   "  unsigned char s[XXH3_SECRET_DEFAULT_SIZE]; XXH3_generateSecret_fromSeed(s, seed);\n"
                   "  XXH128_hash_t a = XXH3_128bits_withSecret(p, n, s, sizeof(s)), b = XXH3_128bits_withSeed(p, n, seed);\n"
                   "  return a.low64 == b.low64 && a.high64 == b.high64; }\n" % os.path.join(kv.ROOT, "third_party", "xxhash", "xxhash.h"))
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", str(tmp_path / "x.so"), str(src)])
    x = ctypes.CDLL(str(tmp_path / "x.so"))
    x.h64.restype = ctypes.c_ulonglong
    x.h64.argtypes = [ctypes.c_char_p, ctypes.c_ulong]
    x.h128.argtypes = [ctypes.c_char_p, ctypes.c_ulong, ctypes.POINTER(ctypes.c_ulonglong)]
    x.secret_matches_seed.argtypes = [ctypes.c_char_p, ctypes.c_ulong, ctypes.c_ulonglong]
    assert x.h64(b"", 0) == 0x2D06800538D394C2
    o = (ctypes.c_ulonglong * 2)()
    x.h128(b"", 0, o)
    assert (o[0], o[1]) == (0x99AA06D3014798D8, 0x6001C324468D497F)
    long_input = bytes(range(256)) * 3
    assert x.secret_matches_seed(long_input, len(long_input), 42) == 1
    assert x.secret_matches_seed(long_input, 241, 42) == 1


def test_store_sort_with_a_smaller_team_than_asked_for(tmp_path):
    """OpenMP may hand out FEWER threads than num_threads() asks for (OMP_THREAD_LIMIT, a caller's own parallel region around the
    call): the bucketed key sort used to cut its shares by the number it asked for, so entries were left out (found by building
    the store writer with the pragmas ignored for the AddressSanitizer run, tests/simt).  8 threads wanted, 2 allowed."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "from stract_amd import _lib\n"
        "from tests import speedy_kv_reader as kv\n"
        "rng = np.random.default_rng(3)\n"
        "n = 200000\n"
        "ids = np.zeros(n, dtype=_lib.U128)\n"
        "ids['lo'] = rng.permutation(n).astype(np.uint64) * 2654435761 + 7\n"
        "ids['hi'] = rng.integers(0, 1 << 40, n, dtype=np.uint64)\n"
        "vals = rng.random(n)\n"
        "_lib.store_write(sys.argv[1], ids, vals)\n"
        "db = kv.Db(sys.argv[1], 'f64', sys.argv[2])\n"
        "ints = kv.ids_to_ints(ids)\n"
        "assert len(db) == n\n"
        "assert all(db.get(ints[j]) == float(vals[j]) for j in rng.integers(0, n, 300).tolist())\n"
        "print('ok')\n")
    env = dict(os.environ, OMP_THREAD_LIMIT="2", HB_HOST_THREADS="8", PYTHONPATH=kv.ROOT)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "db"), str(tmp_path)], capture_output=True, text=True, env=env, cwd=kv.ROOT, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
