"""Independent Python READER of a speedy_kv database directory - TEST INFRASTRUCTURE for include/hb_store.h.

Restates how the reference opens and queries such a directory (paths under /root/reference/crates):
  Db::open_or_create / get / iter      speedy-kv/src/lib.rs:234-262,330-358,510-540
  Segment::open / get_raw / iter_raw   speedy-kv/src/segment.rs:110-131,228-252
  BlobIndex (RandomLookup<BlobPointer>) speedy-kv/src/blob_index.rs:25-58, file-store/src/random_lookup.rs:85-112, lib.rs:46-92
  BlobStore::get_raw                   speedy-kv/src/blob_store.rs:60-83
  BytesBloomFilter::contains_raw       bloom/src/lib.rs:155-166
and the formats of the crates that are not vendored there, written down independently of the writer in
stract_amd/csrc/hb_store.cpp (same published descriptions, different code):
  fst 0.4.7 map (raw/mod.rs Fst::new, raw/node.rs Node::new for the three state kinds), bincode 2 `standard()` integers,
  bitvec 1.0.1's serde form of BitVec<usize, Lsb0>.
"Format unpinned": nothing here was checked against a file written by the reference."""
import ctypes
import json
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LARGE_PRIME = 11400714819323198549
MASK64 = (1 << 64) - 1


# ---- bincode standard(): variable-length integers ---------------------------------------------------------------------
def varint_decode(b, at=0):
    """-> (value, next position)"""
    m = b[at]
    if m < 251:
        return m, at + 1
    n = {251: 2, 252: 4, 253: 8, 254: 16}[m]
    return int.from_bytes(b[at + 1:at + 1 + n], "little"), at + 1 + n


def varint_encode(v):
    if v < 251:
        return bytes([v])
    for marker, n in ((251, 2), (252, 4), (253, 8), (254, 16)):
        if v < 1 << (8 * n):
            return bytes([marker]) + v.to_bytes(n, "little")
    raise ValueError(v)


# ---- CRC-32C (fst footer) ----------------------------------------------------------------------------------------------
def crc32c(data):
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    crc = 0xFFFFFFFF
    for chunk in (data[i:i + (1 << 16)] for i in range(0, len(data), 1 << 16)):
        for x in chunk:
            crc = table[(crc ^ x) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


# ---- fst map -----------------------------------------------------------------------------------------------------------
class FstMap:
    """Reader of an fst 0.4 map file.  A node's address is the position of its state byte; its fields lie below it."""

    def __init__(self, data, verify=False):
        self.d = data
        if len(data) < 36:
            raise ValueError("fst: too small")
        self.version, self.ty = struct.unpack_from("<QQ", data, 0)
        if self.version != 3:
            raise ValueError("fst: version %d" % self.version)
        end = len(data) - 4
        (self.checksum,) = struct.unpack_from("<I", data, end)
        self.len, self.root = struct.unpack_from("<QQ", data, end - 16)
        if self.root + 17 + 4 != len(data):  # Fst::new: the root node is the last node written
            raise ValueError("fst: root address does not end the node section")
        if verify:
            crc = crc32c(data[:end])
            if (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF != self.checksum:
                raise ValueError("fst: checksum mismatch")

    def _le(self, at, n):
        return int.from_bytes(self.d[at:at + n], "little") if n else 0

    def node(self, addr):
        """-> (is_final, final_output, [(input, output, target address)] ascending by input)"""
        if addr == 0:  # EMPTY_ADDRESS: final, no transitions, no output
            return True, 0, []
        d = self.d
        state = d[addr]
        kind = state >> 6
        if kind == 3:  # one transition to the node written just before this one
            if state & 0x3F:
                raise NotImplementedError("fst: common-input table")
            end = addr - 1
            return False, 0, [(d[addr - 1], 0, end - 1)]
        if kind == 2:  # one transition
            if state & 0x3F:
                raise NotImplementedError("fst: common-input table")
            sizes = d[addr - 2]
            tsize, osize = sizes >> 4, sizes & 15
            at = addr - 2 - tsize
            end = at - osize
            delta = self._le(at, tsize)
            return False, 0, [(d[addr - 1], self._le(at - osize, osize), 0 if delta == 0 else end - delta)]
        final = bool(state & 0x40)
        n, nlen = state & 0x3F, 0
        if n == 0:
            n, nlen = d[addr - 1], 1
            if n == 1:
                n = 256
        sizes = d[addr - nlen - 1]
        tsize, osize = sizes >> 4, sizes & 15
        top = addr - nlen - 1 - (256 if n > 32 else 0)
        t_top = top - n
        o_top = t_top - n * tsize
        end = o_top - n * osize - (osize if final else 0)
        trans = []
        for i in range(n):
            delta = self._le(t_top - (i + 1) * tsize, tsize)
            trans.append((d[top - 1 - i], self._le(o_top - (i + 1) * osize, osize), 0 if delta == 0 else end - delta))
        if n > 32:  # the index must agree with the inputs
            for i, (inp, _, _) in enumerate(trans):
                assert d[addr - nlen - 1 - 256 + inp] == (i & 0xFF)
        return final, (self._le(end, osize) if final else 0), trans

    def get(self, key):
        addr, out = self.root, 0
        for byte in key:
            _, _, trans = self.node(addr)
            for inp, o, target in trans:
                if inp == byte:
                    out += o
                    addr = target
                    break
            else:
                return None
        final, fout, _ = self.node(addr)
        return out + fout if final else None

    def items(self):
        """(key bytes, value) in key order"""
        stack = [(self.root, b"", 0)]
        while stack:
            addr, key, out = stack.pop()
            final, fout, trans = self.node(addr)
            if final:
                yield key, out + fout
            for inp, o, target in reversed(trans):
                stack.append((target, key + bytes([inp]), out + o))


# ---- xxh3-128 with the secret derived from seed 42 (bloom/src/lib.rs:27-34), through the xxHash library itself ----------
_shim = None


def _xxh3(tmpdir):
    global _shim
    if _shim is None:
        src = os.path.join(tmpdir, "xxh3_shim.c")
        lib = os.path.join(tmpdir, "xxh3_shim.so")
        with open(src, "w") as f:
            f.write('#define XXH_INLINE_ALL\n#include "%s"\n'
                    "void hash42(const void *p, unsigned long n, unsigned long long *out) {\n"
                    "  static unsigned char secret[XXH3_SECRET_DEFAULT_SIZE]; static int init = 0;\n"
                    "  if (!init) { XXH3_generateSecret_fromSeed(secret, 42); init = 1; }\n"
                    "  XXH128_hash_t h = XXH3_128bits_withSecret(p, n, secret, sizeof(secret)); out[0] = h.high64; out[1] = h.low64; }\n"
                    % os.path.join(ROOT, "third_party", "xxhash", "xxhash.h"))
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", lib, src])
        _shim = ctypes.CDLL(lib)
        _shim.hash42.argtypes = [ctypes.c_char_p, ctypes.c_ulong, ctypes.POINTER(ctypes.c_ulonglong)]
    return _shim


class Bloom:
    def __init__(self, raw, tmpdir):
        n, at = varint_decode(raw)
        self.order = raw[at:at + n].decode()
        at += n
        self.head_width, self.head_index = raw[at], raw[at + 1]
        self.bits, at = varint_decode(raw, at + 2)
        nwords, at = varint_decode(raw, at)
        words = []
        for _ in range(nwords):
            w, at = varint_decode(raw, at)
            words.append(w)
        self.words = words
        self.num_hashes, at = varint_decode(raw, at)
        assert at == len(raw), "trailing bytes in the bloom file"
        assert self.order == "bitvec::order::Lsb0" and self.head_width == 64 and self.head_index == 0
        assert nwords == (self.bits + 63) // 64
        self.tmpdir = tmpdir

    def contains(self, key):
        out = (ctypes.c_ulonglong * 2)()
        _xxh3(self.tmpdir).hash42(key, len(key), out)
        a, b = out[0], out[1]
        for i in range(self.num_hashes):
            h = (((a * i) & MASK64) + b & MASK64) % LARGE_PRIME % self.bits
            if not (self.words[h >> 6] >> (h & 63)) & 1:
                return False
        return True


class Segment:
    def __init__(self, folder, uuid, tmpdir):
        rd = lambda ext: open(os.path.join(folder, uuid + ext), "rb").read()
        self.fst = FstMap(rd(".ids"), verify=True)
        self.bid = rd(".bid")
        self.blobs = rd(".blobs")
        self.bloom = Bloom(rd(".blm"), tmpdir)
        assert len(self.bid) % 32 == 0

    def pointer(self, blob_id):
        return struct.unpack_from("<QQQQ", self.bid, 32 * blob_id)

    def get_raw(self, key):
        if not self.bloom.contains(key):
            return None
        blob_id = self.fst.get(key)
        if blob_id is None:
            return None
        ks, ke, vs, ve = self.pointer(blob_id)
        assert self.blobs[ks:ke] == key
        return self.blobs[vs:ve]

    def iter_raw(self):
        for i in range(len(self.bid) // 32):
            ks, ke, vs, ve = self.pointer(i)
            yield self.blobs[ks:ke], self.blobs[vs:ve]


class Db:
    """kind: 'f64' or 'u64' values; keys are NodeIDs (u128)."""

    def __init__(self, folder, kind, tmpdir):
        meta = json.load(open(os.path.join(folder, "meta.json")))
        assert list(meta) == ["segments"]
        self.segments = [Segment(folder, u, tmpdir) for u in meta["segments"]]
        self.kind = kind

    def _value(self, raw):
        if self.kind == "f64":
            assert len(raw) == 8
            return struct.unpack("<d", raw)[0]
        v, at = varint_decode(raw)
        assert at == len(raw)
        return v

    def __len__(self):
        return sum(s.fst.len for s in self.segments)

    def get(self, node_id):
        key = varint_encode(node_id)
        for s in reversed(self.segments):  # Db::get_raw_with_live: newest segment first (lib.rs:347-353)
            raw = s.get_raw(key)
            if raw is not None:
                return self._value(raw)
        return None

    def items(self):
        for s in self.segments:
            for k, v in s.iter_raw():
                node_id, at = varint_decode(k)
                assert at == len(k)
                yield node_id, self._value(v)


def ids_to_ints(ids):
    return [(int(h) << 64) | int(l) for l, h in zip(ids["lo"].tolist(), ids["hi"].tolist())]


def ints_to_ids(ints, dtype):
    a = np.zeros(len(ints), dtype=dtype)
    a["lo"] = [v & MASK64 for v in ints]
    a["hi"] = [v >> 64 for v in ints]
    return a
