// hb_store.cpp - native writer of speedy_kv databases (include/hb_store.h): what store_harmonic
// (crates/core/src/webgraph/centrality/mod.rs:72-114) leaves on disk, written straight from the result arrays.
// Host only.  Every on-disk format is cited where it is produced; the three that live in un-vendored crates (fst, bitvec's
// serde form, bincode's integer encoding) are restated from their published formats - see the header: FORMAT UNPINNED.
#include <algorithm>
#include <array>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <random>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "../../include/hb_store.h"

#define XXH_INLINE_ALL
#include "../../third_party/xxhash/xxhash.h"

namespace {

using bytes = std::vector<uint8_t>;

// ---- bincode 2.0.0-rc.3, config::standard(): little endian, variable-length integers ------------------------------
// u < 251: one byte; < 2^16: 251 + u16; < 2^32: 252 + u32; < 2^64: 253 + u64; else 254 + u128 (all little endian).
// Floats are their IEEE bytes, little endian.  (crates/common/src/lib.rs:1-3 selects standard().)
size_t varint_u128(unsigned __int128 v, uint8_t *out)
{
    if (v < 251) {
        out[0] = (uint8_t)v;
        return 1;
    }
    int n;
    if (v < ((unsigned __int128)1 << 16)) out[0] = 251, n = 2;
    else if (v < ((unsigned __int128)1 << 32)) out[0] = 252, n = 4;
    else if (v < ((unsigned __int128)1 << 64)) out[0] = 253, n = 8;
    else out[0] = 254, n = 16;
    for (int i = 0; i < n; i++) out[1 + i] = (uint8_t)(v >> (8 * i));
    return (size_t)n + 1;
}
void put_varint(bytes &b, uint64_t v)
{
    uint8_t tmp[17];
    const size_t n = varint_u128(v, tmp);
    b.insert(b.end(), tmp, tmp + n);
}

// ---- buffered file with the running CRC-32C that the fst footer wants ----------------------------------------------
struct Crc32c {
    uint32_t table[256];
    Crc32c()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1; // Castagnoli, reflected
            table[i] = c;
        }
    }
    uint32_t update(uint32_t crc, const uint8_t *p, size_t n) const
    {
        crc = ~crc;
        for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
        return ~crc;
    }
};

class OutFile {
public:
    OutFile(const std::string &path, bool with_crc = false) : path_(path), with_crc_(with_crc)
    {
        f_ = std::fopen(path.c_str(), "wb");
        buf_.reserve(1 << 20);
    }
    ~OutFile()
    {
        if (f_) std::fclose(f_);
    }
    bool ok() const { return f_ != nullptr && !failed_; }
    const std::string &path() const { return path_; }
    uint64_t count() const { return count_; }
    uint32_t crc() const { return crc_; }
    void write(const uint8_t *p, size_t n)
    {
        if (with_crc_) crc_ = crc_tab().update(crc_, p, n);
        count_ += n;
        if (buf_.size() + n > (1u << 20)) flush();
        if (n > (1u << 20)) raw(p, n);
        else buf_.insert(buf_.end(), p, p + n);
    }
    void u8(uint8_t v)
    {
        if (with_crc_ || buf_.size() + 1 > (1u << 20)) return write(&v, 1);
        count_++;
        buf_.push_back(v);
    }
    void le(uint64_t v, int nbytes)
    {
        uint8_t t[8];
        for (int i = 0; i < nbytes; i++) t[i] = (uint8_t)(v >> (8 * i));
        write(t, (size_t)nbytes);
    }
    bool close()
    {
        flush();
        if (f_ && std::fclose(f_) != 0) failed_ = true;
        f_ = nullptr;
        return !failed_;
    }

private:
    static const Crc32c &crc_tab()
    {
        static const Crc32c t;
        return t;
    }
    void raw(const uint8_t *p, size_t n)
    {
        if (f_ && n && std::fwrite(p, 1, n, f_) != n) failed_ = true;
    }
    void flush()
    {
        raw(buf_.data(), buf_.size());
        buf_.clear();
    }
    std::string path_;
    std::FILE *f_ = nullptr;
    bytes buf_;
    uint64_t count_ = 0;
    uint32_t crc_ = 0;
    bool with_crc_, failed_ = false;
};

// ---- fst 0.4.7 map file (crate `fst`, src/raw/{mod,build,node}.rs; format version 3) --------------------------------
// header: version u64 = 3, type u64 = 0.  Nodes are written bottom-up; a node's ADDRESS is the position of its last
// byte (the state byte) and its fields are laid out so that a reader walks backwards from there.  footer: number of
// keys u64, root address u64, masked CRC-32C u32 of everything before it.  Address 0 = the final state without
// transitions and without output (never written); 1 = "no address yet".
// The reference's builder shares equal suffixes through a registry; a reader does not care, so this builder writes the
// plain prefix tree (keys arrive sorted; the values here are 0, 1, 2, ... in key order, which keeps every partial output
// non-negative: a key's value minus the outputs already on the path it shares with its predecessor goes on its first
// own transition).
class FstMapWriter {
public:
    explicit FstMapWriter(OutFile &out) : w_(out)
    {
        w_.le(3, 8); // VERSION
        w_.le(0, 8); // FstType
        push();
    }
    // keys strictly ascending (byte order); value >= every earlier value
    bool insert(const uint8_t *key, size_t len, uint64_t value)
    {
        if (len_ && !(prev_.size() == len ? std::memcmp(prev_.data(), key, len) < 0
                                         : std::lexicographical_compare(prev_.begin(), prev_.end(), key, key + len)))
            return false; // out of order / duplicate
        // common prefix with the unfinished path, and the output already committed along it
        size_t p = 0;
        uint64_t committed = 0;
        while (p < len && p + 1 < depth_ && stack_[p].last_inp == key[p]) {
            committed += stack_[p].last_out;
            p++;
        }
        if (committed > value) return false;
        freeze(p);
        if (p == len) { // the key ends on an existing node (a key that is a prefix of nothing written yet cannot get here)
            stack_[p].node.is_final = true;
            stack_[p].node.final_output = value - committed;
        } else {
            stack_[p].has_last = true;
            stack_[p].last_inp = key[p];
            stack_[p].last_out = value - committed;
            for (size_t i = p + 1; i < len; i++) {
                Unfinished &u = push();
                u.has_last = true;
                u.last_inp = key[i];
            }
            push().node.is_final = true; // the leaf
        }
        prev_.assign(key, key + len);
        len_++;
        return true;
    }
    void finish()
    {
        freeze(0);
        const uint64_t root = compile(stack_[0].node);
        w_.le(len_, 8);
        w_.le(root, 8);
        const uint32_t crc = w_.crc();
        w_.le(((crc >> 15) | (crc << 17)) + 0xa282ead8u, 4); // CountingWriter::masked_checksum
    }

private:
    struct Trans {
        uint8_t inp;
        uint64_t out, addr;
    };
    struct Node {
        bool is_final = false;
        uint64_t final_output = 0;
        std::vector<Trans> trans;
    };
    struct Unfinished {
        Node node;
        bool has_last = false;
        uint8_t last_inp = 0;
        uint64_t last_out = 0;
    };
    static int pack_size(uint64_t n)
    {
        int k = 1;
        while (k < 8 && (n >> (8 * k))) k++;
        return k;
    }
    static uint64_t delta(uint64_t node_addr, uint64_t trans_addr) { return trans_addr == 0 ? 0 : node_addr - trans_addr; }

    // keep the first `keep` + 1 unfinished nodes; compile the deeper ones bottom-up, each into its parent's last transition
    void freeze(size_t keep)
    {
        while (depth_ > keep + 1) {
            const uint64_t addr = compile(stack_[depth_ - 1].node);
            depth_--;
            Unfinished &parent = stack_[depth_ - 1];
            parent.node.trans.push_back(Trans{parent.last_inp, parent.last_out, addr});
            parent.has_last = false;
        }
    }
    // the unfinished path lives in a pool that only grows: a popped entry keeps its transition vector's capacity
    Unfinished &push()
    {
        if (depth_ == stack_.size()) stack_.emplace_back();
        Unfinished &u = stack_[depth_++];
        u.node.is_final = false;
        u.node.final_output = 0;
        u.node.trans.clear();
        u.has_last = false;
        u.last_inp = 0;
        u.last_out = 0;
        return u;
    }
    // build.rs Builder::compile + node.rs Node::compile_to
    uint64_t compile(const Node &n)
    {
        if (n.is_final && n.trans.empty() && n.final_output == 0) return 0; // EMPTY_ADDRESS
        const uint64_t start = w_.count();
        if (!n.is_final && n.trans.size() == 1) {
            const Trans &t = n.trans[0];
            if (t.out == 0 && t.addr == last_addr_) { // StateOneTransNext: the target is the node written just before
                w_.u8(t.inp);                           // (common-input index 0: the input byte is stored)
                w_.u8(0xC0);
            } else { // StateOneTrans: [output][target delta][pack sizes][input][state]
                const int osize = t.out ? pack_size(t.out) : 0, tsize = pack_size(delta(start, t.addr));
                if (osize) w_.le(t.out, osize);
                w_.le(delta(start, t.addr), tsize);
                w_.u8((uint8_t)((tsize << 4) | osize));
                w_.u8(t.inp);
                w_.u8(0x80);
            }
        } else { // StateAnyTrans: [final output][outputs, reversed][target deltas, reversed][inputs, reversed][index][pack sizes][count][state]
            int tsize = 0, osize = pack_size(n.final_output);
            bool any_outs = n.final_output != 0;
            for (const Trans &t : n.trans) {
                tsize = std::max(tsize, pack_size(delta(start, t.addr)));
                osize = std::max(osize, pack_size(t.out));
                any_outs = any_outs || t.out != 0;
            }
            if (!any_outs) osize = 0;
            if (any_outs) {
                if (n.is_final) w_.le(n.final_output, osize);
                for (size_t i = n.trans.size(); i-- > 0;) w_.le(n.trans[i].out, osize);
            }
            for (size_t i = n.trans.size(); i-- > 0;) w_.le(delta(start, n.trans[i].addr), tsize);
            for (size_t i = n.trans.size(); i-- > 0;) w_.u8(n.trans[i].inp);
            if (n.trans.size() > 32) { // TRANS_INDEX_THRESHOLD: input byte -> transition number
                uint8_t index[256];
                std::memset(index, 255, sizeof(index));
                for (size_t i = 0; i < n.trans.size(); i++) index[n.trans[i].inp] = (uint8_t)i;
                w_.write(index, 256);
            }
            w_.u8((uint8_t)((tsize << 4) | osize));
            uint8_t state = n.is_final ? 0x40 : 0x00;
            if (n.trans.size() >= 1 && n.trans.size() <= 63) state |= (uint8_t)n.trans.size();
            else w_.u8(n.trans.size() == 256 ? 1 : (uint8_t)n.trans.size());
            w_.u8(state);
        }
        last_addr_ = w_.count() - 1;
        return last_addr_;
    }

    OutFile &w_;
    std::vector<Unfinished> stack_;
    size_t depth_ = 0;
    bytes prev_;
    uint64_t len_ = 0, last_addr_ = 1; // NONE_ADDRESS
};

// ---- bloom::BytesBloomFilter (crates/bloom/src/lib.rs:36-48,132-178) -------------------------------------------------
struct Bloom {
    uint64_t num_bits, num_hashes;
    std::vector<uint64_t> words; // BitVec<usize, Lsb0>: bit i = word i / 64, bit i % 64
    uint8_t secret[XXH3_SECRET_DEFAULT_SIZE];
    explicit Bloom(uint64_t estimated_items)
    {
        const double fp = 0.01, ln2 = std::log(2.0);
        num_bits = (uint64_t)std::ceil((double)estimated_items * std::log(fp) / (-8.0 * (ln2 * ln2))); // lib.rs:38-41
        const double h = std::ceil((double)num_bits / (double)estimated_items * ln2);                     // lib.rs:45-48
        num_hashes = std::max<uint64_t>((h >= 0.0 && h == h) ? (uint64_t)h : 0, 1);                        // `as u64` saturates, NaN -> 0
        words.assign((size_t)((num_bits + 63) / 64), 0);
        XXH3_generateSecret_fromSeed(secret, 42); // = xxhash_rust::const_xxh3::const_custom_default_secret(42), lib.rs:27
    }
    void insert(const uint8_t *key, size_t len)
    {
        if (!num_bits) return; // `% 0` would panic in the reference: an empty database writes no segment at all
        const XXH128_hash_t h = XXH3_128bits_withSecret(key, len, secret, sizeof(secret));
        const uint64_t a = h.high64, b = h.low64; // split_u128: [high, low]
        for (uint64_t i = 0; i < num_hashes; i++) {
            const uint64_t x = ((a * i + b) % 11400714819323198549ull) % num_bits; // lib.rs:171-176
            words[(size_t)(x >> 6)] |= 1ull << (x & 63);
        }
    }
    // bincode of { #[bincode(with_serde)] bit_vec: BitVec, num_hashes: u64, PhantomData }.  bitvec 1.0.1 serialises a bit
    // sequence as the struct { order: type_name::<Lsb0>(), head: BitIdx { width: u8, index: u8 }, bits: u64, data: [usize] }
    // (bitvec/src/serdes/slice.rs); through bincode's serde bridge: strings and sequences carry a variable-length length,
    // u8 is one byte, u64 / usize are variable-length integers.
    void serialize(bytes &out) const
    {
        static const char order[] = "bitvec::order::Lsb0";
        put_varint(out, sizeof(order) - 1);
        out.insert(out.end(), order, order + sizeof(order) - 1);
        out.push_back(64); // head.width: bits of usize
        out.push_back(0);  // head.index: BitVec::repeat starts at bit 0
        put_varint(out, num_bits);
        put_varint(out, words.size());
        for (uint64_t w : words) put_varint(out, w);
        put_varint(out, num_hashes);
    }
};

struct Entry {
    std::array<uint8_t, 17> key; // bincode(NodeID), zero padded (the first byte fixes the length: padding never decides an order)
    uint8_t key_len;
    uint64_t index; // into the caller's arrays
};

std::string uuid_v4()
{
    std::random_device rd;
    uint8_t b[16];
    for (int i = 0; i < 16; i += 4) {
        const uint32_t r = rd();
        std::memcpy(b + i, &r, 4);
    }
    b[6] = (uint8_t)((b[6] & 0x0F) | 0x40); // version 4
    b[8] = (uint8_t)((b[8] & 0x3F) | 0x80); // RFC 4122 variant
    char s[37];
    std::snprintf(s, sizeof(s), "%02x%02x%02x%02x-%02x%02x-%02x%02x-%02x%02x-%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4], b[5], b[6],
                  b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15]);
    return s;
}

int fail(char *err, size_t err_len, int code, const std::string &msg)
{
    if (err && err_len) std::snprintf(err, err_len, "%s", msg.c_str());
    return code;
}

bool make_dirs(const std::string &path)
{
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && ::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

int write_db(const char *dir_c, const hb_u128 *ids, const void *values, int value_kind, uint64_t count, char *err, size_t err_len)
{
    if (!dir_c || !*dir_c) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: dir is empty");
    if (count && (!ids || !values)) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: NULL array with count > 0");
    if (value_kind != HB_STORE_F64 && value_kind != HB_STORE_U64) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: unknown value kind");
    const std::string dir(dir_c);
    if (!make_dirs(dir)) return fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot create directory " + dir);
    const std::string meta_path = dir + "/meta.json";
    if (count == 0) { // Db::commit with an empty live segment writes nothing (lib.rs:376-379); open_or_create saved an empty Meta
        OutFile meta(meta_path);
        static const char empty[] = "{\n  \"segments\": []\n}";
        meta.write((const uint8_t *)empty, sizeof(empty) - 1);
        return meta.close() ? HB_OK : fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot write " + meta_path);
    }
    // keys in ascending byte order of their encodings (LiveSegment is a BTreeMap<Vec<u8>, _>, lib.rs:108-110)
    std::vector<Entry> entries(count);
    for (uint64_t i = 0; i < count; i++) {
        Entry &e = entries[i];
        e.key.fill(0);
        e.key_len = (uint8_t)varint_u128(((unsigned __int128)ids[i].hi << 64) | ids[i].lo, e.key.data());
        e.index = i;
    }
    std::sort(entries.begin(), entries.end(), [](const Entry &a, const Entry &b) { return std::memcmp(a.key.data(), b.key.data(), 17) < 0; });
    for (uint64_t i = 1; i < count; i++)
        if (entries[i].key == entries[i - 1].key) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: duplicate NodeID");

    const std::string uuid = uuid_v4(), base = dir + "/" + uuid;
    OutFile blobs(base + ".blobs"), bid(base + ".bid"), idsf(base + ".ids", true), blm(base + ".blm");
    for (OutFile *f : {&blobs, &bid, &idsf, &blm})
        if (!f->ok()) return fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot create " + f->path());
    FstMapWriter fst(idsf);
    Bloom bloom(count); // SegmentWriter::new(num_items, ..): BytesBloomFilter::new(num_items, 0.01), segment.rs:56-59
    uint64_t offset = 0;
    for (uint64_t i = 0; i < count; i++) { // SegmentWriter::insert, segment.rs:66-75
        const Entry &e = entries[i];
        uint8_t val[9];
        size_t val_len;
        if (value_kind == HB_STORE_F64) {
            std::memcpy(val, &((const double *)values)[e.index], 8); // little-endian host (gfx950 boxes are x86-64)
            val_len = 8;
        } else {
            uint8_t tmp[17];
            val_len = varint_u128(((const uint64_t *)values)[e.index], tmp);
            std::memcpy(val, tmp, val_len);
        }
        blobs.write(e.key.data(), e.key_len);
        blobs.write(val, val_len);
        bid.le(offset, 8); // BlobPointer: key range, value range
        bid.le(offset + e.key_len, 8);
        bid.le(offset + e.key_len, 8);
        bid.le(offset + e.key_len + val_len, 8);
        offset += e.key_len + val_len;
        if (!fst.insert(e.key.data(), e.key_len, i)) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: keys not strictly ascending");
        bloom.insert(e.key.data(), e.key_len);
    }
    fst.finish();
    bytes b;
    bloom.serialize(b);
    blm.write(b.data(), b.size());
    for (OutFile *f : {&blobs, &bid, &idsf, &blm})
        if (!f->close()) return fail(err, err_len, HB_ERR_IO, "hb_store_write: write failed on " + f->path());
    // Meta { segments: [uuid] } through serde_json::to_string_pretty (lib.rs:292-297)
    OutFile meta(meta_path);
    const std::string js = "{\n  \"segments\": [\n    \"" + uuid + "\"\n  ]\n}";
    meta.write((const uint8_t *)js.data(), js.size());
    if (!meta.close()) return fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot write " + meta_path);
    return HB_OK;
}

} // namespace

extern "C" int hb_store_write(const char *dir, const hb_u128 *ids, const void *values, int value_kind, uint64_t count, char *err, size_t err_len)
{
    if (err && err_len) err[0] = 0;
    try {
        return write_db(dir, ids, values, value_kind, count, err, err_len);
    } catch (const std::bad_alloc &) {
        return fail(err, err_len, HB_ERR_NOMEM, "hb_store_write: out of host memory");
    } catch (const std::exception &e) {
        return fail(err, err_len, HB_ERR_INVALID, std::string("hb_store_write: ") + e.what());
    }
}

extern "C" int hb_store_harmonic(const char *output, const hb_u128 *ids, const double *centralities, const uint64_t *ranks, uint64_t count,
                                 char *err, size_t err_len)
{
    if (!output || !*output) return fail(err, err_len, HB_ERR_INVALID, "hb_store_harmonic: output is empty");
    const std::string out(output);
    int rc = hb_store_write((out + "/harmonic").c_str(), ids, centralities, HB_STORE_F64, count, err, err_len);
    if (rc != HB_OK) return rc;
    return hb_store_write((out + "/harmonic_rank").c_str(), ids, ranks, HB_STORE_U64, count, err, err_len);
}
