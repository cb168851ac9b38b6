// hb_webgraph.cpp - native reader of Stract's webgraph edge store (include/hb_webgraph.h states the format and
// cites the reference serialisers it follows).  Host C++ only; nothing is copied from the reference.
#include "../../include/hb_webgraph.h"
#include "hb_threads.h"

#include <immintrin.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_open_error;

struct Bytes {
    const uint8_t *p = nullptr;
    uint64_t n = 0;
    Bytes() = default;
    Bytes(const uint8_t *p_, uint64_t n_) : p(p_), n(n_) {}
    Bytes sub(uint64_t off, uint64_t len) const { return Bytes(p + off, len); }
};

uint32_t rd_u32(const uint8_t *p)
{
    uint32_t v;
    std::memcpy(&v, p, 4);
    return v; // little-endian host (x86-64): the format is little-endian throughout (common/serialize.rs Endianness)
}
uint64_t rd_u64(const uint8_t *p)
{
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}

// ---- CRC-32 (IEEE 802.3, the polynomial of crc32fast), slicing-by-8 ------------------------------------
struct Crc32Tables {
    uint32_t t[8][256];
    Crc32Tables()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
// state -> state over n bytes (no pre / post inversion), table form
uint32_t crc32_tables(uint32_t c, const uint8_t *p, uint64_t n)
{
    static const Crc32Tables T;
    while (n >= 8) {
        const uint32_t a = rd_u32(p) ^ c, b = rd_u32(p + 4);
        c = T.t[7][a & 0xFF] ^ T.t[6][(a >> 8) & 0xFF] ^ T.t[5][(a >> 16) & 0xFF] ^ T.t[4][a >> 24] ^ T.t[3][b & 0xFF] ^
            T.t[2][(b >> 8) & 0xFF] ^ T.t[1][(b >> 16) & 0xFF] ^ T.t[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}
// [r6] the same state update by carry-less multiplication (PCLMULQDQ): four 16-byte lanes are folded 64 bytes ahead per step, then
// into one, then reduced to 32 bits (Barrett) - Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ
// Instruction" (Intel, 2009); constants for the bit-reflected IEEE polynomial 0xEDB88320: x^(4*128+64), x^(4*128), x^(128+64), x^128,
// x^64 mod P, then P' and mu.  ~8x the table form per core: the store check of hb_load_webgraph was CPU-bound on the cores the
// process may use (20 GB/s on 16 CPUs).  n >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t c, const uint8_t *p, uint64_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(p + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(p + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(p + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(p + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    x0 = k1k2;
    p += 64;
    n -= 64;
    while (n >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(p + 0x00));
        y6 = _mm_loadu_si128((const __m128i *)(p + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(p + 0x20));
        y8 = _mm_loadu_si128((const __m128i *)(p + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        p += 64;
        n -= 64;
    }
    x0 = k3k4; // the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (n >= 16) { // single 16-byte folds
        x2 = _mm_loadu_si128((const __m128i *)p);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        p += 16;
        n -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction to 32 bits
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
bool crc32_have_clmul()
{
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !std::getenv("HB_CRC32_TABLES"); // (env: the table form, for A/B runs)
    return have;
}
uint32_t crc32_ieee(const uint8_t *p, uint64_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    if (n >= 64 && crc32_have_clmul()) {
        const uint64_t body = n & ~15ull;
        c = crc32_clmul(c, p, body);
        p += body;
        n -= body;
    }
    return crc32_tables(c, p, n) ^ 0xFFFFFFFFu;
}
// crc of the concatenation A ++ B from crc(A), crc(B) and |B| (the zlib construction: crc(A) is advanced over |B| zero
// bytes by repeated squaring of the "one zero bit" operator over GF(2), then xor-ed with crc(B))
uint32_t gf2_times(const uint32_t *mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (; vec; vec >>= 1, mat++)
        if (vec & 1) sum ^= *mat;
    return sum;
}
void gf2_square(uint32_t *sq, const uint32_t *mat)
{
    for (int k = 0; k < 32; k++) sq[k] = gf2_times(mat, mat[k]);
}
uint32_t crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    if (!len2) return crc1;
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u; // operator for one zero bit
    for (int k = 1; k < 32; k++) odd[k] = 1u << (k - 1);
    gf2_square(even, odd); // two zero bits
    gf2_square(odd, even); // four
    do {                   // first squaring below: one zero byte
        gf2_square(even, odd);
        if (len2 & 1) crc1 = gf2_times(even, crc1);
        len2 >>= 1;
        if (!len2) break;
        gf2_square(odd, even);
        if (len2 & 1) crc1 = gf2_times(odd, crc1);
        len2 >>= 1;
    } while (len2);
    return crc1 ^ crc2;
}
// the same CRC over a large range, pieces in parallel (a 100 GB store at one thread's 1.5 GB/s would take a minute)
uint32_t crc32_ieee_parallel(const uint8_t *p, uint64_t n)
{
    const uint64_t piece = 64ull << 20;
    if (n <= 2 * piece) return crc32_ieee(p, n);
    const uint64_t pieces = (n + piece - 1) / piece;
    std::vector<uint32_t> part((size_t)pieces);
#pragma omp parallel for num_threads(hb::host_threads()) schedule(dynamic, 1)
    for (int64_t k = 0; k < (int64_t)pieces; k++) part[(size_t)k] = crc32_ieee(p + (uint64_t)k * piece, std::min(piece, n - (uint64_t)k * piece));
    uint32_t c = part[0];
    for (uint64_t k = 1; k < pieces; k++) c = crc32_combine(c, part[(size_t)k], std::min(piece, n - k * piece));
    return c;
}

// ---- zstd, only if a dictionary block is compressed (blocks above 2048 bytes, sstable/delta.rs:58-80) --------
typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef unsigned long long (*zstd_bound_fn)(const void *, size_t);
typedef unsigned (*zstd_iserr_fn)(size_t);
bool zstd_decompress(Bytes in, std::vector<uint8_t> *out, std::string *err)
{
    static void *lib = nullptr;
    static zstd_decompress_fn dec = nullptr;
    static zstd_bound_fn bound = nullptr;
    static zstd_iserr_fn iserr = nullptr;
    if (!lib) {
        for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (lib) {
            dec = (zstd_decompress_fn)dlsym(lib, "ZSTD_decompress");
            bound = (zstd_bound_fn)dlsym(lib, "ZSTD_getFrameContentSize");
            iserr = (zstd_iserr_fn)dlsym(lib, "ZSTD_isError");
        }
    }
    if (!dec || !bound || !iserr) {
        *err = "a dictionary block is zstd-compressed and libzstd.so.1 cannot be loaded";
        return false;
    }
    unsigned long long sz = bound(in.p, in.n);
    if (sz == 0xFFFFFFFFFFFFFFFFull || sz == 0xFFFFFFFFFFFFFFFEull) sz = 1024 * 1024; // unknown: the reference's fallback
    out->resize((size_t)sz);
    const size_t got = dec(out->data(), out->size(), in.p, in.n);
    if (iserr(got)) {
        *err = "zstd: corrupt dictionary block";
        return false;
    }
    out->resize(got);
    return true;
}

// ---- sstable (sstable/mod.rs, delta.rs, block_reader.rs, vint.rs, value/range.rs) --------------------------
// LEB128: continue bit on every byte but the last (sstable/vint.rs - NOT common/vint.rs, whose stop bit is inverted)
bool sst_vint(Bytes *b, uint64_t *out)
{
    uint64_t r = 0;
    int shift = 0;
    while (b->n) {
        const uint8_t c = *b->p;
        b->p++;
        b->n--;
        r |= (uint64_t)(c & 127u) << shift;
        if (c < 128) {
            *out = r;
            return true;
        }
        shift += 7;
        if (shift > 63) return false;
    }
    return false;
}

struct SstEntry {
    std::string key;
    uint64_t start = 0, end = 0;
};

// value_mode 0: VoidSSTable (no value block); 1: RangeSSTable
std::string read_sstable(Bytes dict, int value_mode, std::vector<SstEntry> *out)
{
    if (dict.n < 20) return "sstable shorter than its footer";
    const uint64_t index_offset = rd_u64(dict.p + dict.n - 20);
    const uint64_t num_terms = rd_u64(dict.p + dict.n - 12);
    const uint32_t version = rd_u32(dict.p + dict.n - 4);
    if (version != 2 && version != 3) return "unsupported sstable version " + std::to_string(version);
    if (index_offset > dict.n - 20) return "sstable: index offset out of range";
    Bytes blocks = dict.sub(0, index_offset); // the block index (fst + block addresses) is not needed to stream all entries
    std::vector<uint8_t> scratch;
    std::string key;
    while (blocks.n) {
        if (blocks.n < 4) return "sstable: truncated block length";
        uint64_t block_len = rd_u32(blocks.p);
        blocks = blocks.sub(4, blocks.n - 4);
        if (block_len <= 1) break; // end marker
        if (blocks.n < block_len) return "sstable: truncated block";
        const uint8_t compress = blocks.p[0];
        Bytes body = blocks.sub(1, block_len - 1);
        blocks = blocks.sub(block_len, blocks.n - block_len);
        if (compress == 1) {
            std::string err;
            if (!zstd_decompress(body, &scratch, &err)) return err;
            body = Bytes(scratch.data(), scratch.size());
        } else if (compress != 0) {
            return "sstable: unknown block compression flag";
        }
        // values of the block first (value/range.rs: count, first start, then end - start deltas)
        std::vector<uint64_t> bounds;
        if (value_mode == 1) {
            uint64_t cnt = 0;
            if (!sst_vint(&body, &cnt)) return "sstable: bad value block";
            uint64_t prev = 0;
            for (uint64_t i = 0; i < cnt; i++) {
                uint64_t d = 0;
                if (!sst_vint(&body, &d)) return "sstable: bad value block";
                prev += d;
                bounds.push_back(prev);
            }
        }
        // keys: (keep, add) packed in one byte when both < 16, else 0x01 followed by two vints (delta.rs:90-100,160-180)
        key.clear(); // the first key of a block is stored whole (the writer clears previous_key per block)
        uint64_t idx = 0;
        while (body.n) {
            const uint8_t b0 = body.p[0];
            body = body.sub(1, body.n - 1);
            uint64_t keep, add;
            if (b0 == 1) {
                if (!sst_vint(&body, &keep) || !sst_vint(&body, &add)) return "sstable: bad key header";
            } else {
                keep = b0 & 15u;
                add = b0 >> 4;
            }
            if (keep > key.size() || add > body.n) return "sstable: bad key delta";
            key.resize(keep);
            key.append((const char *)body.p, add);
            body = body.sub(add, body.n - add);
            SstEntry e;
            e.key = key;
            if (value_mode == 1) {
                if (idx + 1 >= bounds.size()) return "sstable: fewer values than keys";
                e.start = bounds[idx];
                e.end = bounds[idx + 1];
            }
            out->push_back(e);
            idx++;
        }
    }
    if (out->size() != num_terms) return "sstable: term count mismatch";
    return "";
}

// ---- minimal JSON (meta.json) ---------------------------------------------------------------------------------
struct Json {
    const char *p, *e;
    std::string err;
    void ws()
    {
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    bool lit(char c)
    {
        ws();
        if (p < e && *p == c) {
            p++;
            return true;
        }
        return false;
    }
    bool str(std::string *out)
    {
        ws();
        if (p >= e || *p != '"') return false;
        p++;
        out->clear();
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                p++;
                switch (*p) {
                case 'n': out->push_back('\n'); break;
                case 't': out->push_back('\t'); break;
                case 'u': p += std::min<long>(4, (long)(e - p - 1)); out->push_back('?'); break;
                default: out->push_back(*p);
                }
                p++;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= e) return false;
        p++;
        return true;
    }
    bool skip() // any value
    {
        ws();
        if (p >= e) return false;
        if (*p == '"') {
            std::string s;
            return str(&s);
        }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            p++;
            if (lit(close)) return true;
            while (true) {
                if (open == '{') {
                    std::string k;
                    if (!str(&k) || !lit(':')) return false;
                }
                if (!skip()) return false;
                if (lit(',')) continue;
                return lit(close);
            }
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n') p++; // number / true / false / null
        return true;
    }
    bool number(uint64_t *out)
    {
        ws();
        if (p >= e || *p < '0' || *p > '9') return false;
        uint64_t v = 0;
        while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (uint64_t)(*p++ - '0');
        *out = v;
        return true;
    }
};

struct SegmentMeta {
    std::string uuid; // 32 hex chars
    uint64_t max_doc = 0;
};

std::string parse_meta(const std::string &text, std::vector<SegmentMeta> *segs)
{
    Json j{text.data(), text.data() + text.size(), ""};
    if (!j.lit('{')) return "meta.json: not an object";
    bool found = false;
    if (!j.lit('}')) {
        while (true) {
            std::string k;
            if (!j.str(&k) || !j.lit(':')) return "meta.json: malformed";
            if (k == "segments") {
                found = true;
                if (!j.lit('[')) return "meta.json: \"segments\" is not an array";
                if (!j.lit(']')) {
                    while (true) {
                        if (!j.lit('{')) return "meta.json: segment entry is not an object";
                        SegmentMeta s;
                        bool have_id = false, have_docs = false;
                        if (!j.lit('}')) {
                            while (true) {
                                std::string sk;
                                if (!j.str(&sk) || !j.lit(':')) return "meta.json: malformed segment entry";
                                if (sk == "segment_id") {
                                    std::string id;
                                    if (!j.str(&id)) return "meta.json: segment_id is not a string";
                                    for (char ch : id)
                                        if (ch != '-') s.uuid.push_back((char)std::tolower((unsigned char)ch)); // serde writes the hyphenated form
                                    have_id = true;
                                } else if (sk == "max_doc") {
                                    if (!j.number(&s.max_doc)) return "meta.json: max_doc is not a number";
                                    have_docs = true;
                                } else if (!j.skip()) {
                                    return "meta.json: malformed segment entry";
                                }
                                if (j.lit(',')) continue;
                                if (!j.lit('}')) return "meta.json: malformed segment entry";
                                break;
                            }
                        }
                        if (!have_id || !have_docs || s.uuid.size() != 32) return "meta.json: segment entry without segment_id / max_doc";
                        segs->push_back(s);
                        if (j.lit(',')) continue;
                        if (!j.lit(']')) return "meta.json: malformed segment list";
                        break;
                    }
                }
            } else if (!j.skip()) {
                return "meta.json: malformed";
            }
            if (j.lit(',')) continue;
            if (!j.lit('}')) return "meta.json: malformed";
            break;
        }
    }
    if (!found) return "meta.json: no \"segments\"";
    return "";
}

// ---- one segment ------------------------------------------------------------------------------------------------
struct Segment {
    SegmentMeta meta;
    void *map = nullptr;
    uint64_t map_len = 0;
    const uint8_t *from = nullptr, *to = nullptr, *flags = nullptr; // raw little-endian values, num_rows each
    const uint8_t *page_from = nullptr, *page_to = nullptr;         // `from_id` / `to_id` (HBW_PAGE_IDS)
    uint64_t num_rows = 0;
    uint64_t first = 0; // stream position of its first document
    uint64_t body_len = 0;   // bytes the directory footer's CRC-32 covers
    uint64_t crc_want = 0;   // ... and its value (crc_known: the footer carries one)
    bool crc_known = false;
};

// column bytes -> raw value array (column/serialize.rs:27-59)
std::string open_raw_column(Bytes col, int value_bytes, uint8_t want_codec, uint64_t expect_rows, const uint8_t **data)
{
    if (col.n < 4) return "column shorter than its trailer";
    const uint32_t index_bytes = rd_u32(col.p + col.n - 4);
    Bytes body = col.sub(0, col.n - 4);
    if (index_bytes < 1 || index_bytes > body.n) return "column index length out of range";
    if (body.p[0] != 0) return "column cardinality is not Full (code " + std::to_string(body.p[0]) + ")";
    Bytes vals = body.sub(index_bytes, body.n - index_bytes);
    const uint64_t header = 1 + 4 + 2 * (uint64_t)value_bytes; // codec, num_rows, min, max
    if (vals.n < header) return "column values shorter than their header";
    if (vals.p[0] != want_codec)
        return "column codec " + std::to_string(vals.p[0]) + " is not Raw (" + std::to_string(want_codec) +
               "): the reference snapshot only writes Raw (column/serialize.rs:23,53)";
    const uint64_t rows = rd_u32(vals.p + 1);
    if (rows != expect_rows) return "column holds " + std::to_string(rows) + " rows, the segment " + std::to_string(expect_rows);
    if (vals.n - header < rows * (uint64_t)value_bytes) return "column data truncated";
    *data = vals.p + header;
    return "";
}

std::string open_segment(const std::string &dir, uint32_t flags, Segment *s)
{
    const std::string path = dir + "/" + s->meta.uuid + ".col";
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return path + ": cannot open";
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) {
        ::close(fd);
        return path + ": too small";
    }
    s->map_len = (uint64_t)st.st_size;
    s->map = mmap(nullptr, s->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (s->map == MAP_FAILED) {
        s->map = nullptr;
        return path + ": mmap failed";
    }
    Bytes file((const uint8_t *)s->map, s->map_len);
    // directory footer (footer.rs:45-102): ... [json][json_len u32][1337 u32]
    const uint32_t magic = rd_u32(file.p + file.n - 4), json_len = rd_u32(file.p + file.n - 8);
    if (magic != 1337) return path + ": footer magic mismatch";
    if (json_len > 50000 || (uint64_t)json_len + 8 > file.n) return path + ": footer length out of range";
    Bytes body = file.sub(0, file.n - 8 - json_len);
    {
        const std::string js((const char *)file.p + body.n, json_len);
        const size_t at = js.find("\"crc\":");
        s->body_len = body.n;
        s->crc_known = at != std::string::npos;
        if (s->crc_known) s->crc_want = std::strtoull(js.c_str() + at + 6, nullptr, 10);
    }
    if (flags & HBW_VERIFY_CRC) {
        if (!s->crc_known) return path + ": footer without crc";
        if ((uint64_t)crc32_ieee_parallel(body.p, body.n) != s->crc_want) return path + ": CRC mismatch";
    }
    // columnar footer (reader/mod.rs:85-103): [column data][sstable][sstable_len u64][num_rows u32][version u32][magic 4]
    if (body.n < 20) return path + ": columnar body too small";
    const uint8_t *tail = body.p + body.n - 20;
    const uint64_t sstable_len = rd_u64(tail);
    const uint32_t num_rows = rd_u32(tail + 8), version = rd_u32(tail + 12);
    static const uint8_t kMagic[4] = {2, 113, 119, 66};
    if (std::memcmp(tail + 16, kMagic, 4) != 0) return path + ": not a columnar file";
    if (version != 1) return path + ": unsupported columnar version";
    if (sstable_len > body.n - 20) return path + ": dictionary length out of range";
    if (num_rows != s->meta.max_doc) return path + ": num_rows differs from meta.json max_doc";
    const Bytes column_data = body.sub(0, body.n - 20 - sstable_len);
    const Bytes dict = body.sub(column_data.n, sstable_len);
    std::vector<SstEntry> cols;
    std::string e = read_sstable(dict, 1, &cols);
    if (!e.empty()) return path + ": " + e;
    s->num_rows = num_rows;
    struct Want {
        const char *name;
        uint8_t type_code; // column_type.rs: U64 = 1, U128 = 6
        int value_bytes;
        uint8_t codec;     // Raw: 0 among the u128 codecs, 3 among the u64 codecs
        const uint8_t **dst;
    } wants[5] = {{"from_host_id", 6, 16, 0, &s->from}, {"to_host_id", 6, 16, 0, &s->to}, {"rel_flags", 1, 8, 3, &s->flags},
                  {"from_id", 6, 16, 0, &s->page_from}, {"to_id", 6, 16, 0, &s->page_to}}; // schema.rs:132-180: page node ids
    const int nwants = (flags & HBW_PAGE_IDS) ? 5 : 3;
    for (int wi = 0; wi < nwants; wi++) {
        const Want &w = wants[wi];
        std::string key(w.name);
        key.push_back('\0');
        key.push_back((char)w.type_code);
        const SstEntry *hit = nullptr;
        for (const SstEntry &c : cols)
            if (c.key == key) hit = &c;
        if (!hit) return path + ": no column `" + w.name + "` of the expected type";
        if (hit->end < hit->start || hit->end > column_data.n) return path + ": column range out of bounds";
        e = open_raw_column(column_data.sub(hit->start, hit->end - hit->start), w.value_bytes, w.codec, num_rows, w.dst);
        if (!e.empty()) return path + ": `" + w.name + "`: " + e;
    }
    return "";
}

} // namespace

struct hbw_reader {
    std::string dir;
    std::vector<Segment> segs;
    uint64_t total = 0;
    std::string err;
    ~hbw_reader()
    {
        for (Segment &s : segs)
            if (s.map) {
                // in pieces [r6]: one munmap of a multi-GB mapping holds the process' address-space lock for as long as its page
                // tables take to tear down (C4: 101 GB, 3.3 s in all), and everything else in the process that needs that lock in
                // the meantime - a thread being created, a large malloc - waits: the first hb_finish behind hb_load_webgraph took
                // 419 ms against 4.7 ms.  64 MiB per call = ~2 ms per hold.
                const size_t piece = 64u << 20;
                for (size_t off = 0; off < s.map_len; off += piece) munmap((char *)s.map + off, std::min(piece, s.map_len - off));
            }
    }
};

namespace {
template <class F>
int guarded(hbw_reader *r, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        try { (r ? r->err : g_open_error) = "out of host memory"; } catch (...) {}
        return HB_ERR_NOMEM;
    } catch (...) {
        try { (r ? r->err : g_open_error) = "unexpected C++ exception"; } catch (...) {}
        return HB_ERR_INVALID;
    }
}
} // namespace

extern "C" {

const char *hbw_last_error(const hbw_reader *r) { return r ? r->err.c_str() : g_open_error.c_str(); }

int hbw_open(const char *edges_dir, uint32_t flags, hbw_reader **out)
{
    return guarded(nullptr, [&]() -> int {
        if (!edges_dir || !out) {
            g_open_error = "NULL argument";
            return HB_ERR_INVALID;
        }
        *out = nullptr;
        const std::string dir(edges_dir);
        std::string text;
        {
            FILE *f = std::fopen((dir + "/meta.json").c_str(), "rb");
            if (!f) {
                g_open_error = dir + "/meta.json: cannot open";
                return HB_ERR_INVALID;
            }
            char buf[65536];
            size_t got;
            while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
            std::fclose(f);
        }
        hbw_reader *r = new hbw_reader();
        r->dir = dir;
        std::vector<SegmentMeta> metas;
        std::string e = parse_meta(text, &metas);
        for (size_t i = 0; e.empty() && i < metas.size(); i++) {
            r->segs.emplace_back();
            Segment &s = r->segs.back();
            s.meta = metas[i];
            s.first = r->total;
            e = open_segment(dir, flags, &s);
            r->total += s.num_rows;
        }
        if (!e.empty()) {
            g_open_error = e;
            delete r;
            return HB_ERR_INVALID;
        }
        *out = r;
        return HB_OK;
    });
}

void hbw_close(hbw_reader *r) { delete r; }

int hbw_num_segments(const hbw_reader *r, uint64_t *count)
{
    if (!r || !count) return HB_ERR_INVALID;
    *count = r->segs.size();
    return HB_OK;
}

int hbw_segment_info(const hbw_reader *r, uint64_t segment, char uuid[33], uint64_t *num_rows)
{
    if (!r || segment >= r->segs.size()) return HB_ERR_INVALID;
    if (uuid) std::snprintf(uuid, 33, "%s", r->segs[segment].meta.uuid.c_str());
    if (num_rows) *num_rows = r->segs[segment].num_rows;
    return HB_OK;
}

int hbw_total_rows(const hbw_reader *r, uint64_t *rows)
{
    if (!r || !rows) return HB_ERR_INVALID;
    *rows = r->total;
    return HB_OK;
}

static int read_records(const hbw_reader *r, bool page_level, uint64_t first, uint64_t count, hb_edge *out)
{
    if (!r || (count && !out)) return HB_ERR_INVALID;
    if (first > r->total || count > r->total - first) return HB_ERR_INVALID;
    uint64_t done = 0;
    for (const Segment &s : r->segs) {
        if (done == count) break;
        const uint64_t pos = first + done;
        if (pos >= s.first + s.num_rows) continue;
        const uint8_t *from = page_level ? s.page_from : s.from, *to = page_level ? s.page_to : s.to;
        if (!from || !to) return HB_ERR_INVALID; // page-level ids: the reader was opened without HBW_PAGE_IDS
        const uint64_t b = pos - s.first, n = std::min<uint64_t>(count - done, s.num_rows - b);
        hb_edge *dst = out + done;
#pragma omp parallel for num_threads(hb::host_threads()) schedule(static) if (n > 65536)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            std::memcpy(&dst[i].from, from + 16 * (b + (uint64_t)i), 16); // u128 little-endian = {lo, hi}
            std::memcpy(&dst[i].to, to + 16 * (b + (uint64_t)i), 16);
            std::memcpy(&dst[i].rel_flags, s.flags + 8 * (b + (uint64_t)i), 8);
        }
        done += n;
    }
    return HB_OK;
}

int hbw_read_host_edges(const hbw_reader *r, uint64_t first, uint64_t count, hb_edge *out) { return read_records(r, false, first, count, out); }

int hbw_read_page_edges(const hbw_reader *r, uint64_t first, uint64_t count, hb_edge *out) { return read_records(r, true, first, count, out); }

int hb_load_webgraph(hb_ctx *ctx, const char *edges_dir, uint32_t flags)
{
    if (!ctx) return HB_ERR_INVALID;
    hbw_reader *raw = nullptr;
    const bool trace = std::getenv("HB_TRACE_INGEST") != nullptr; // where the time of this call goes, on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    // HBW_VERIFY_CRC: the files are NOT checked in a pass of their own here (hbw_open does that for its direct users) - the
    // check runs on a second thread WHILE the records stream to the device (the column bytes are read once, the cores idle
    // while the device reduces); its verdict is taken before hb_finalize, a mismatch discards what was appended
    int rc = hbw_open(edges_dir, flags & ~(uint32_t)HBW_VERIFY_CRC, &raw);
    if (rc != HB_OK) return rc; // message: hbw_last_error(NULL)
    const double s_open = since(t_begin);
    // closed on every path, exceptions included - on a thread of its own: unmapping a 100 GB store takes seconds (3.3 of 9.4 s
    // at C4, profiles/r04l_bench_C4_load_webgraph_trace.txt) and nobody has to wait for it
    struct CloseInBackground {
        void operator()(hbw_reader *p) const
        {
            if (!p) return;
            try {
                std::thread([p]() { hbw_close(p); }).detach();
            } catch (...) {
                hbw_close(p);
            }
        }
    };
    std::unique_ptr<hbw_reader, CloseInBackground> guard(raw);
    hbw_reader *r = raw;
    return guarded(r, [&]() -> int {
        // 4 Mi records (160 MiB) per hand-over.  HB_WEBGRAPH_SLAB_RECORDS (tests): a small slab makes a small store take the
        // many-slab path - the alternating pinned buffers, the reader thread waiting for a free one, this thread for a full one
        uint64_t slab = 1ull << 22;
        if (const char *e = std::getenv("HB_WEBGRAPH_SLAB_RECORDS")) {
            const uint64_t v = std::strtoull(e, nullptr, 10);
            if (v >= 1 && v <= (1ull << 24)) slab = v;
        }
        if (flags & HBW_PAGE_IDS) {
            // page-level records need a context in reference-tail mode: find out before the whole store is ingested
            int rc0 = hb_load_tail_edges(ctx, nullptr, 0);
            const char *msg = hb_last_error(ctx);
            if (rc0 != HB_OK && msg && std::strstr(msg, "HB_FLAG_REFERENCE_TAIL")) {
                g_open_error = msg;
                return rc0;
            }
        }
        // two pinned slabs: a reader thread gathers slab k + 1 out of the mapped column files (all cores, read_records)
        // while this thread hands slab k to the library (H2D at link rate + the table kernels)
        struct Pinned {
            hb_edge *p[2] = {nullptr, nullptr};
            ~Pinned()
            {
                for (hb_edge *q : p) hb_pinned_free(q);
            }
        } pin;
        const uint64_t cap = std::min<uint64_t>(slab, std::max<uint64_t>(r->total, 1));
        for (int k = 0; k < 2; k++) {
            void *q = nullptr;
            if (hb_pinned_alloc(cap * sizeof(hb_edge), &q) != HB_OK) {
                r->err = "hb_load_webgraph: cannot allocate the pinned record slabs";
                g_open_error = r->err; // (the reader is freed on return: hbw_last_error(NULL) keeps the text)
                return HB_ERR_NOMEM;
            }
            pin.p[k] = (hb_edge *)q;
        }
        hb_edge *buf = pin.p[0]; // (the page-level pass below reuses the first slab)
        const uint64_t nslabs = (r->total + slab - 1) / slab;
        std::mutex mu;
        std::condition_variable cv;
        uint64_t filled = 0, taken = 0; // slabs gathered / handed over
        int rd_rc = HB_OK;
        bool stop = false;
        double s_gather = 0, s_append = 0, s_wait = 0; // reader thread busy / this thread in hb_append_edges / waiting for a slab
        std::string crc_error;
        double s_crc = 0;
        std::thread checker;
        if (flags & HBW_VERIFY_CRC)
            checker = std::thread([&]() {
                const auto t_c = std::chrono::steady_clock::now();
                for (const Segment &sg : r->segs) {
                    if (!crc_error.empty()) break;
                    const std::string path = r->dir + "/" + sg.meta.uuid + ".col";
                    if (!sg.crc_known) crc_error = path + ": footer without crc";
                    else if ((uint64_t)crc32_ieee_parallel((const uint8_t *)sg.map, sg.body_len) != sg.crc_want) crc_error = path + ": CRC mismatch";
                }
                s_crc = since(t_c);
            });
        struct JoinChecker {
            std::thread &t;
            ~JoinChecker()
            {
                if (t.joinable()) t.join();
            }
        } join_checker{checker};
        std::thread producer([&]() {
            for (uint64_t k = 0; k < nslabs; k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k < taken + 2; }); // its buffer is free again
                    if (stop) return;
                }
                const uint64_t at = k * slab, n = std::min(slab, r->total - at);
                const auto t_g = std::chrono::steady_clock::now();
                const int rc = hbw_read_host_edges(r, at, n, pin.p[k & 1]);
                s_gather += since(t_g);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (rc != HB_OK) rd_rc = rc;
                    filled = k + 1;
                }
                cv.notify_all();
                if (rc != HB_OK) return;
            }
        });
        // whatever leaves this scope - a return, an exception out of hb_append_edges' guard - first stops and joins the reader
        // thread (a joinable std::thread that is destroyed calls std::terminate, past every catch handler)
        struct JoinProducer {
            std::thread &t;
            std::mutex &mu;
            std::condition_variable &cv;
            bool &stop;
            ~JoinProducer()
            {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    stop = true;
                }
                cv.notify_all();
                if (t.joinable()) t.join();
            }
        } join_producer{producer, mu, cv, stop};
        int rc2 = HB_OK;
        for (uint64_t k = 0; k < nslabs && rc2 == HB_OK; k++) {
            {
                const auto t_w = std::chrono::steady_clock::now();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return filled > k || rd_rc != HB_OK; });
                if (filled <= k) rc2 = rd_rc;
                s_wait += since(t_w);
            }
            const auto t_a = std::chrono::steady_clock::now();
            if (rc2 == HB_OK) rc2 = hb_append_edges(ctx, pin.p[k & 1], std::min(slab, r->total - k * slab));
            s_append += since(t_a);
            {
                std::lock_guard<std::mutex> lk(mu);
                taken = k + 1;
                if (rc2 != HB_OK) stop = true;
            }
            cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = stop || rc2 != HB_OK;
        }
        cv.notify_all();
        producer.join();
        if (checker.joinable()) checker.join();
        if (rc2 == HB_OK && !crc_error.empty()) { // a damaged file: nothing of it may become a graph
            (void)hb_discard_appended(ctx);
            r->err = crc_error;
            g_open_error = crc_error;
            return HB_ERR_INVALID;
        }
        const auto t_f = std::chrono::steady_clock::now();
        if (rc2 == HB_OK) rc2 = hb_finalize(ctx, nullptr, 0); // node set = all endpoints = host_nodes() (store.rs:338-357)
        if (trace)
            std::fprintf(stderr, "[hb webgraph] %llu records in %zu segments: open %.3f s; %llu slabs: reader thread gathering %.3f s, this thread waiting "
                                 "for a slab %.3f s, in hb_append_edges %.3f s; CRC-32 of every file on a second thread %.3f s; hb_finalize %.3f s; total "
                                 "%.3f s\n", (unsigned long long)r->total, r->segs.size(), s_open, (unsigned long long)nslabs, s_gather, s_wait, s_append, s_crc,
                         since(t_f), since(t_begin));
        if (rc2 == HB_OK && (flags & HBW_PAGE_IDS)) {
            // HB_FLAG_REFERENCE_TAIL: every document's page-level (from_id, to_id, rel_flags), segment by segment in doc
            // order (a ForwardlinksQuery runs one LinksScorer per segment; its de-duplication depends on that order,
            // query/raw/links.rs:115-232); the library keeps what ForwardlinksQuery::new(host id) can return (harmonic.rs:82-92)
            rc2 = hb_load_tail_edges(ctx, nullptr, 0);
            for (const Segment &s : r->segs) {
                for (uint64_t at = 0; at < s.num_rows && rc2 == HB_OK; at += slab) {
                    const uint64_t n = std::min(slab, s.num_rows - at);
                    rc2 = hbw_read_page_edges(r, s.first + at, n, buf);
                    if (rc2 == HB_OK) rc2 = hb_append_tail_edges(ctx, buf, n);
                }
                if (rc2 == HB_OK) rc2 = hb_tail_segment_end(ctx);
            }
        }
        return rc2;
    });
}

int hbw_debug_sstable(const uint8_t *bytes, uint64_t len, int value_mode, uint8_t *keys_out, uint64_t keys_cap, uint64_t *ranges_out,
                      uint64_t ranges_cap, uint64_t *count)
{
    return guarded(nullptr, [&]() -> int {
        if (!bytes || !count) return HB_ERR_INVALID;
        std::vector<SstEntry> ents;
        const std::string e = read_sstable(Bytes(bytes, len), value_mode, &ents);
        if (!e.empty()) {
            g_open_error = e;
            return HB_ERR_INVALID;
        }
        *count = ents.size();
        uint64_t kpos = 0;
        for (size_t i = 0; i < ents.size(); i++) {
            const uint32_t kl = (uint32_t)ents[i].key.size();
            if (keys_out && kpos + 4 + kl <= keys_cap) {
                std::memcpy(keys_out + kpos, &kl, 4);
                std::memcpy(keys_out + kpos + 4, ents[i].key.data(), kl);
            }
            kpos += 4 + kl;
            if (ranges_out && 2 * i + 1 < ranges_cap) {
                ranges_out[2 * i] = ents[i].start;
                ranges_out[2 * i + 1] = ents[i].end;
            }
        }
        return HB_OK;
    });
}

uint32_t hbw_debug_crc32(const uint8_t *bytes, uint64_t len) { return crc32_ieee_parallel(bytes, len); } // (pieces + combine above 128 MiB)

} // extern "C"
