// hb_store.cpp - native writer of speedy_kv databases (include/hb_store.h): what store_harmonic
// (crates/core/src/webgraph/centrality/mod.rs:72-114) leaves on disk, written straight from the result arrays.
// Host only.  Every on-disk format is cited where it is produced; the three that live in un-vendored crates (fst, bitvec's
// serde form, bincode's integer encoding) are restated from their published formats - see the header: FORMAT UNPINNED.
//
// Round 4: every stage runs on all host cores (OpenMP) - round 3 was one thread, 1 M entries/s:
//   keys      bincode(NodeID) per entry, ordered by a parallel multiway merge sort of 24-byte (key words, index) entries
//   .blobs    \ byte offsets by a prefix sum over the sorted entries, then every thread formats its block of entries and
//   .bid      / writes it at its file offset (pwrite)
//   .blm      bloom inserts with atomic ORs, one XXH3-128 per key
//   .ids      the fst map is built as INDEPENDENT SUB-TRIES, one per 3-byte key prefix, in parallel: node addresses inside
//             an fst are stored as distances (node address - target address), so a sub-trie's bytes do not depend on where
//             it lands in the file; the sequential part only writes the top three levels and concatenates.  The result is
//             byte-identical to feeding all keys through one builder (HB_STORE_FST=sequential keeps that path; the tests
//             compare the two files).
//   hb_store_harmonic: both databases share keys, order, bloom filter and fst - sorted and built once, written twice.
#include <algorithm>
#include <array>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <new>
#include <random>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <fcntl.h>
#include <nmmintrin.h>
#include <omp.h>
#include <parallel/algorithm>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/hb_store.h"
#include "hb_internal.h"
#include "hb_threads.h"

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

// CRC-32C (Castagnoli, reflected) with the SSE4.2 instruction: the fst footer carries it over the whole file
__attribute__((target("sse4.2"))) uint32_t crc32c_update(uint32_t crc, const uint8_t *p, size_t n)
{
    uint64_t c = (uint32_t)~crc;
    while (n && ((uintptr_t)p & 7)) {
        c = _mm_crc32_u8((uint32_t)c, *p++);
        n--;
    }
    for (; n >= 8; n -= 8, p += 8) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        c = _mm_crc32_u64(c, v);
    }
    while (n--) c = _mm_crc32_u8((uint32_t)c, *p++);
    return ~(uint32_t)c;
}

// crc of A ++ B from crc(A), crc(B) and |B| - the zlib construction (crc(A) advanced over |B| zero bytes by repeated squaring of the
// "one zero bit" operator over GF(2)) for the reflected Castagnoli polynomial; lets the fst writer take the CRC of a sub-trie that
// another thread computed and account for it in the running file CRC without touching the bytes again
uint32_t gf2_times32(const uint32_t *mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (; vec; vec >>= 1, mat++)
        if (vec & 1) sum ^= *mat;
    return sum;
}
void gf2_square32(uint32_t *sq, const uint32_t *mat)
{
    for (int k = 0; k < 32; k++) sq[k] = gf2_times32(mat, mat[k]);
}
// [r6] the operators for 2^k zero BYTES are squared up once per process (64 matrices of 32 words); a combine is then one matrix-vector
// product per set bit of len2 (~0.1 us) instead of ~15 matrix squarings (~10 us): the sequential top of the parallel fst build calls it
// once per sub-trie - 65 536 times for a store of hashed NodeIDs, which alone cost more than building a 1 M-key fst on one thread.
struct Crc32cPowers {
    uint32_t m[64][32];
    Crc32cPowers()
    {
        uint32_t even[32], odd[32];
        odd[0] = 0x82F63B78u; // operator for one zero bit
        for (int k = 1; k < 32; k++) odd[k] = 1u << (k - 1);
        gf2_square32(even, odd);  // two zero bits
        gf2_square32(odd, even);  // four
        gf2_square32(m[0], odd);  // eight = one zero byte
        for (int k = 1; k < 64; k++) gf2_square32(m[k], m[k - 1]);
    }
};
uint32_t crc32c_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    static const Crc32cPowers pw; // (thread-safe initialisation)
    if (!len2) return crc1;
    for (int k = 0; len2; len2 >>= 1, k++)
        if (len2 & 1) crc1 = gf2_times32(pw.m[k], crc1);
    return crc1 ^ crc2;
}

bool pwrite_all(int fd, const uint8_t *p, size_t n, uint64_t off);

// ---- byte sinks of the fst writer -----------------------------------------------------------------------------------
// a file with the running byte count and CRC.  Small writes are buffered and go out with pwrite at their offset; a BLOCK whose CRC the
// caller already has (a finished sub-trie, [r6]) is only booked - offset, pointer, length; running CRC by crc32c_combine - and written
// by close() with all other blocks on all host cores: the sequential top of the parallel fst build no longer copies and checksums
// 1.3 GB (C4) on one thread.  The blocks must stay alive until close().
class FileSink {
public:
    explicit FileSink(const std::string &path) : path_(path)
    {
        fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd_ < 0) failed_ = true;
        buf_.reserve(1 << 20);
    }
    ~FileSink()
    {
        if (fd_ >= 0) ::close(fd_);
    }
    bool ok() const { return !failed_; }
    const std::string &path() const { return path_; }
    uint64_t count() const { return count_; }
    uint32_t crc() const { return crc_; }
    void write(const uint8_t *p, size_t n)
    {
        crc_ = crc32c_update(crc_, p, n);
        count_ += n;
        if (buf_.size() + n > (1u << 20)) flush();
        if (n > (1u << 20)) raw(p, n);
        else buf_.insert(buf_.end(), p, p + n);
    }
    void write_block(const uint8_t *p, size_t n, uint32_t crc_of_block)
    {
        if (!n) return;
        // what is buffered in front of the block is booked too (kept in owned_), not written: no system call in the sequential phase
        if (!buf_.empty()) {
            owned_.emplace_back();
            owned_.back().swap(buf_);
            blocks_.push_back(Block{pos_, owned_.back().data(), owned_.back().size()});
            buf_.reserve(1 << 12);
        }
        blocks_.push_back(Block{count_, p, n});
        crc_ = crc32c_combine(crc_, crc_of_block, n);
        count_ += n;
        pos_ = count_;
    }
    void u8(uint8_t v) { write(&v, 1); }
    void le(uint64_t v, int nbytes)
    {
        uint8_t t[8];
        for (int i = 0; i < nbytes; i++) t[i] = (uint8_t)(v >> (8 * i));
        write(t, (size_t)nbytes);
    }
    bool close()
    {
        flush();
        if (fd_ >= 0 && !blocks_.empty()) {
            // The booked blocks are adjacent in file order.  Writes to ONE file queue behind its inode lock whatever the number of writers,
            // and a store of hashed NodeIDs books 131 072 blocks of a few KB: one pwrite each was 131 072 system calls taking turns.  Runs
            // of >= 8 MB are gathered into a staging buffer by the team (the copies run in parallel) and go out as one write each.
            std::atomic<bool> ok{true};
            const int fd = fd_;
            const std::vector<Block> &bl = blocks_;
            std::vector<size_t> cut{0};
            {
                uint64_t run = 0;
                for (size_t k = 0; k < bl.size(); k++) {
                    const bool adjacent = k && bl[k - 1].off + bl[k - 1].n == bl[k].off;
                    if (k && (!adjacent || run >= (8u << 20))) {
                        cut.push_back(k);
                        run = 0;
                    }
                    run += bl[k].n;
                }
                cut.push_back(bl.size());
            }
            const size_t nruns = cut.size() - 1;
#pragma omp parallel num_threads(hb::host_threads())
            {
                bytes stage;
#pragma omp for schedule(dynamic, 1)
                for (size_t c = 0; c < nruns; c++) {
                    const size_t a = cut[c], b = cut[c + 1];
                    if (b == a + 1) {
                        if (!pwrite_all(fd, bl[a].p, bl[a].n, bl[a].off)) ok = false;
                        continue;
                    }
                    const uint64_t total = bl[b - 1].off + bl[b - 1].n - bl[a].off;
                    stage.resize(total);
                    for (size_t k = a; k < b; k++) std::memcpy(stage.data() + (bl[k].off - bl[a].off), bl[k].p, bl[k].n);
                    if (!pwrite_all(fd, stage.data(), total, bl[a].off)) ok = false;
                }
            }
            if (!ok) failed_ = true;
            blocks_.clear();
            owned_.clear();
        }
        if (fd_ >= 0 && ::close(fd_) != 0) failed_ = true;
        fd_ = -1;
        return !failed_;
    }

private:
    struct Block {
        uint64_t off;
        const uint8_t *p;
        size_t n;
    };
    void raw(const uint8_t *p, size_t n)
    {
        if (fd_ < 0) {
            failed_ = true;
            return;
        }
        if (n && !pwrite_all(fd_, p, n, pos_)) failed_ = true;
        pos_ += n;
    }
    void flush()
    {
        raw(buf_.data(), buf_.size());
        buf_.clear();
    }
    std::string path_;
    int fd_ = -1;
    bytes buf_;
    std::vector<Block> blocks_;
    std::deque<bytes> owned_; // small buffered runs between two booked blocks (a deque: their addresses stay put)
    uint64_t count_ = 0; // bytes accepted so far = file offset of the next byte
    uint64_t pos_ = 0;   // file offset of the first byte still in buf_
    uint32_t crc_ = 0;
    bool failed_ = false;
};

// memory, with a virtual position: a sub-trie is compiled as if it started at file offset kSubBase, so that its node
// addresses never collide with the two reserved addresses (0 = the empty final node, 1 = none); only differences of
// addresses are ever stored, so the base cancels
constexpr uint64_t kSubBase = 16;
class MemSink {
public:
    bytes data;
    uint64_t count() const { return kSubBase + data.size(); }
    void write(const uint8_t *p, size_t n) { data.insert(data.end(), p, p + n); }
    void write_block(const uint8_t *p, size_t n, uint32_t) { write(p, n); }
    // n more bytes at the end, to be filled in by the caller (FstWriter::emit_tail)
    uint8_t *grab(size_t n)
    {
        const size_t at = data.size();
        data.resize(at + n);
        return data.data() + at;
    }
    void u8(uint8_t v) { data.push_back(v); }
    void le(uint64_t v, int nbytes)
    {
        for (int i = 0; i < nbytes; i++) data.push_back((uint8_t)(v >> (8 * i)));
    }
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
template <class Sink>
class FstWriter {
public:
    // whole_file: header + footer around the nodes (the .ids file); otherwise a bare sub-trie (finish_sub)
    FstWriter(Sink &out, bool whole_file) : w_(out), whole_(whole_file)
    {
        if (whole_) {
            w_.le(3, 8); // VERSION
            w_.le(0, 8); // FstType
        }
        push();
    }
    // keys strictly ascending (byte order); value >= every earlier value
    bool insert(const uint8_t *key, size_t len, uint64_t value)
    {
        if (len_ && !(prev_.size() == len ? std::memcmp(prev_.data(), key, len) < 0
                                         : std::lexicographical_compare(prev_.begin(), prev_.end(), key, key + len)))
            return false; // out of order / duplicate
        // common prefix with the unfinished path, and the output already committed along it.  [r6] The part of the previous key below its
        // branching point is IMPLICIT (tail_on_): nodes depth_ .. L (L = its length) exist only as the bytes prev_[depth_ .. L) - a chain of
        // one-transition nodes without outputs ending in the final leaf, which is what the tail of a key of a map of hashes always is.  It
        // is written as bytes (emit_tail) when the next key branches off above it, and turned into stack entries (materialize) only as far as
        // the next key follows it; before, every key pushed ~12 stack entries that the next key froze again one by one.
        const size_t L = prev_.size();
        size_t p = 0;
        uint64_t committed = 0;
        if (tail_on_) {
            while (p < len && p < depth_ && stack_[p].last_inp == key[p]) { // (every explicit node has a pending transition here)
                committed += stack_[p].last_out;
                p++;
            }
            if (p == depth_)
                while (p < len && p < L && prev_[p] == key[p]) p++; // along the implicit chain: no outputs
        } else {
            while (p < len && p + 1 < depth_ && stack_[p].last_inp == key[p]) {
                committed += stack_[p].last_out;
                p++;
            }
        }
        if (committed > value) return false;
        if (tail_on_) {
            if (p >= depth_) materialize(p); // the new key follows the chain down to node p
            if (tail_on_) attach(emit_tail(), p); // what hangs below the deepest explicit node is complete
        }
        freeze(p);
        if (p == len) { // the key ends on an existing node (a key that is a prefix of nothing written yet cannot get here)
            stack_[p].node.is_final = true;
            stack_[p].node.final_output = value - committed;
        } else {
            stack_[p].has_last = true;
            stack_[p].last_inp = key[p];
            stack_[p].last_out = value - committed;
            tail_on_ = true; // nodes p + 1 .. len: implicit (depth_ == p + 1)
        }
        prev_.assign(key, key + len);
        len_++;
        return true;
    }
    // A compiled sub-trie (`sub`, built by FstWriter<MemSink> over the key suffixes behind `prefix`, values relative to
    // `value`, `keys` keys, root at virtual address `sub_root`) hangs below the path `prefix`: exactly what insert() of its
    // keys one by one would have produced.  prefix must be > every key inserted so far and no prefix of one.
    bool insert_subtrie(const uint8_t *prefix, size_t plen, uint64_t value, const bytes &sub, uint32_t sub_crc, uint64_t sub_root, uint64_t keys,
                        const uint8_t *last_key, size_t last_len)
    {
        if (!plen || !keys) return false;
        if (len_ && !std::lexicographical_compare(prev_.begin(), prev_.end(), prefix, prefix + plen)) return false;
        if (tail_on_) materialize(prev_.size()); // (rare path: the stack as the code below expects it)
        size_t p = 0;
        uint64_t committed = 0;
        while (p < plen && p + 1 < depth_ && stack_[p].last_inp == prefix[p]) {
            committed += stack_[p].last_out;
            p++;
        }
        if (p == plen || committed > value) return false;
        freeze(p);
        stack_[p].has_last = true;
        stack_[p].last_inp = prefix[p];
        stack_[p].last_out = value - committed;
        for (size_t i = p + 1; i < plen; i++) {
            Unfinished &u = push();
            u.has_last = true;
            u.last_inp = prefix[i];
        }
        // the node behind the prefix is the sub-trie's root: its bytes go out now, the last unfinished node points at it
        const uint64_t base = w_.count();
        w_.write_block(sub.data(), sub.size(), sub_crc); // (a file sink books it and writes it at close(): `sub` must stay alive until then)
        const uint64_t addr = base - kSubBase + sub_root;
        Unfinished &parent = stack_[depth_ - 1];
        parent.node.trans.push_back(Trans{parent.last_inp, parent.last_out, addr});
        parent.has_last = false;
        last_addr_ = addr; // the root is the last node a sub-trie writes
        prev_.assign(last_key, last_key + last_len);
        len_ += keys;
        return true;
    }
    void finish()
    {
        if (tail_on_) attach(emit_tail(), 0);
        freeze(0);
        const uint64_t root = compile(stack_[0].node);
        w_.le(len_, 8);
        w_.le(root, 8);
        const uint32_t crc = w_.crc();
        w_.le(((crc >> 15) | (crc << 17)) + 0xa282ead8u, 4); // CountingWriter::masked_checksum
    }
    // bare sub-trie: the address of its root node (which must have been written: a sub-trie has at least one key of
    // length >= 1, so its root has a transition)
    uint64_t finish_sub()
    {
        if (tail_on_) attach(emit_tail(), 0);
        freeze(0);
        return compile(stack_[0].node);
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
            attach(addr, keep);
        }
    }
    // `addr` = the compiled node behind the pending transition of the deepest unfinished node
    void attach(uint64_t addr, size_t keep)
    {
        Unfinished &parent = stack_[depth_ - 1];
        // a parent that will be frozen in this very call and has nothing but this child (the tail of a key below its
        // branching point: most nodes of a map of hashes) is written without going through its transition list
        if (depth_ > keep + 1 && !parent.node.is_final && parent.node.trans.empty()) {
            const uint64_t a2 = compile_one(parent.last_inp, parent.last_out, addr);
            parent.has_last = false;
            depth_--;
            Unfinished &gp = stack_[depth_ - 1];
            gp.node.trans.push_back(Trans{gp.last_inp, gp.last_out, a2});
            gp.has_last = false;
            return;
        }
        parent.node.trans.push_back(Trans{parent.last_inp, parent.last_out, addr});
        parent.has_last = false;
    }
    // the implicit tail (nodes depth_ .. L of the previous key) as bytes, deepest node first: the leaf is the empty final node (address
    // 0, never written), the node above it a one-transition node pointing at address 0, every node above that a one-transition node
    // whose target is the node written just before it (StateOneTransNext: two bytes).  Returns the address of node depth_.
    uint64_t emit_tail()
    {
        const size_t L = prev_.size();
        tail_on_ = false;
        if (depth_ >= L) return 0; // only the leaf was implicit
        if constexpr (std::is_same<Sink, MemSink>::value) {
            // the same bytes compile_one would write, stored in one go: [0x00 target delta][0x10 pack sizes][input][0x80] for the node above the
            // leaf (StateOneTrans to address 0), then [input][0xC0] (StateOneTransNext) per node above it
            uint8_t *q = w_.grab(4 + 2 * (L - 1 - depth_));
            q[0] = 0x00;
            q[1] = 0x10;
            q[2] = prev_[L - 1];
            q[3] = 0x80;
            q += 4;
            for (size_t d = L - 1; d-- > depth_;) {
                q[0] = prev_[d];
                q[1] = 0xC0;
                q += 2;
            }
            last_addr_ = w_.count() - 1;
            return last_addr_;
        } else {
            uint64_t addr = compile_one(prev_[L - 1], 0, 0);
            for (size_t d = L - 1; d-- > depth_;) addr = compile_one(prev_[d], 0, addr);
            return addr;
        }
    }
    // implicit nodes depth_ .. upto become stack entries (upto <= L; node L is the final leaf)
    void materialize(size_t upto)
    {
        const size_t L = prev_.size();
        while (depth_ <= upto) {
            const size_t d = depth_;
            Unfinished &u = push();
            if (d < L) {
                u.has_last = true;
                u.last_inp = prev_[d];
            } else {
                u.node.is_final = true;
            }
        }
        if (upto >= L) tail_on_ = false;
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
    // a non-final node with exactly one transition (node.rs: StateOneTransNext / StateOneTrans)
    uint64_t compile_one(uint8_t inp, uint64_t out, uint64_t addr)
    {
        const uint64_t start = w_.count();
        if (out == 0 && addr == last_addr_) { // StateOneTransNext: the target is the node written just before
            w_.u8(inp);                         // (common-input index 0: the input byte is stored)
            w_.u8(0xC0);
        } else { // StateOneTrans: [output][target delta][pack sizes][input][state]
            const int osize = out ? pack_size(out) : 0, tsize = pack_size(delta(start, addr));
            if (osize) w_.le(out, osize);
            w_.le(delta(start, addr), tsize);
            w_.u8((uint8_t)((tsize << 4) | osize));
            w_.u8(inp);
            w_.u8(0x80);
        }
        last_addr_ = w_.count() - 1;
        return last_addr_;
    }
    // build.rs Builder::compile + node.rs Node::compile_to
    uint64_t compile(const Node &n)
    {
        if (n.is_final && n.trans.empty() && n.final_output == 0) return 0; // EMPTY_ADDRESS
        if (!n.is_final && n.trans.size() == 1) return compile_one(n.trans[0].inp, n.trans[0].out, n.trans[0].addr);
        const uint64_t start = w_.count();
        // StateAnyTrans: [final output][outputs, reversed][target deltas, reversed][inputs, reversed][index][pack sizes][count][state]
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
        last_addr_ = w_.count() - 1;
        return last_addr_;
    }

    Sink &w_;
    bool whole_;
    std::vector<Unfinished> stack_;
    size_t depth_ = 0;
    bytes prev_;
    uint64_t len_ = 0, last_addr_ = 1; // NONE_ADDRESS
    bool tail_on_ = false;             // the previous key's nodes below depth_ - 1 are implicit (insert)
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
    void insert(const uint8_t *key, size_t len) // safe to call from many threads at once
    {
        if (!num_bits) return; // `% 0` would panic in the reference: an empty database writes no segment at all
        const XXH128_hash_t h = XXH3_128bits_withSecret(key, len, secret, sizeof(secret));
        const uint64_t a = h.high64, b = h.low64; // split_u128: [high, low]
        for (uint64_t i = 0; i < num_hashes; i++) {
            const uint64_t x = ((a * i + b) % 11400714819323198549ull) % num_bits; // lib.rs:171-176
            __atomic_fetch_or(&words[(size_t)(x >> 6)], 1ull << (x & 63), __ATOMIC_RELAXED);
        }
    }
    // [r6] the same inserts for a run of entries, the way the store writer calls it: a key costs num_hashes (7 at fp = 0.01) bit
    // positions, each two 64-bit remainders and one atomic OR somewhere in a ~1.2 byte-per-key bit vector (C4: 95 MB - a cache miss
    // each).  The remainder by num_bits is taken with a precomputed 128-bit reciprocal (Lemire, Kaser, Kurz: "Faster remainder by direct
    // computation", 2019; exact for every 64-bit operand - and checked against `%` when the filter is built), and the positions of
    // kBatch keys are computed and prefetched before any of them is written: 1.1 -> 0.3 s for C4's 79 M keys on 16 threads.
    struct FastMod {
        uint64_t d = 1;
        unsigned __int128 M = 0;
        bool exact = false;
        void init(uint64_t div)
        {
            d = div ? div : 1;
            M = ~(unsigned __int128)0 / d + 1;
            exact = d > 1;
            uint64_t s = 0x9E3779B97F4A7C15ull;
            for (int k = 0; k < 4096 && exact; k++) { // the construction is proven; this guards the transcription
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const uint64_t a = k < 8 ? (uint64_t)0 - (uint64_t)k : s;
                if (fast(a) != a % d) exact = false;
            }
        }
        uint64_t fast(uint64_t a) const
        {
            const unsigned __int128 low = M * a; // mod 2^128
            const unsigned __int128 t = (unsigned __int128)(uint64_t)low * d;
            const unsigned __int128 r = (unsigned __int128)(uint64_t)(low >> 64) * d + (uint64_t)(t >> 64);
            return (uint64_t)(r >> 64);
        }
        uint64_t mod(uint64_t a) const { return exact ? fast(a) : a % d; }
    };
    template <class KeyOf>
    void insert_range(uint64_t lo, uint64_t hi, KeyOf &&key_of) // entries [lo, hi); key_of(i, buf) fills buf and returns the key's length
    {
        if (!num_bits) return;
        FastMod fm;
        fm.init(num_bits);
        constexpr uint64_t kBatch = 16, kMaxHashes = 16;
        if (num_hashes > kMaxHashes) {
            uint8_t key[17];
            for (uint64_t i = lo; i < hi; i++) insert(key, key_of(i, key));
            return;
        }
        uint64_t pos[kBatch * kMaxHashes];
        for (uint64_t base = lo; base < hi; base += kBatch) {
            const uint64_t cnt = std::min(kBatch, hi - base);
            uint64_t np = 0;
            for (uint64_t j = 0; j < cnt; j++) {
                uint8_t key[17];
                const size_t len = key_of(base + j, key);
                const XXH128_hash_t h = XXH3_128bits_withSecret(key, len, secret, sizeof(secret));
                const uint64_t a = h.high64, b = h.low64;
                for (uint64_t i = 0; i < num_hashes; i++) {
                    const uint64_t x = fm.mod((a * i + b) % 11400714819323198549ull);
                    __builtin_prefetch(&words[(size_t)(x >> 6)], 1, 0);
                    pos[np++] = x;
                }
            }
            for (uint64_t k = 0; k < np; k++) __atomic_fetch_or(&words[(size_t)(pos[k] >> 6)], 1ull << (pos[k] & 63), __ATOMIC_RELAXED);
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
        out.reserve(out.size() + words.size() * 9 + 16);
        for (uint64_t w : words) put_varint(out, w);
        put_varint(out, num_hashes);
    }
};

using Entry = hb::StoreKey; // (hb_internal.h: the device sort of hb_store_harmonic_results produces the same records)
using EntryVec = hb::StoreKeyVec; // (elements left uninitialised on resize: every user fills them at once)

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

// <dir>/meta.json through a temporary file + rename: a crash mid-write leaves the old database or none, never a torn file
bool write_meta(const std::string &dir, const std::string &json)
{
    const std::string tmp = dir + "/meta.json.tmp", dst = dir + "/meta.json";
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(json.data(), 1, json.size(), f) == json.size();
    ok = (std::fflush(f) == 0) && ok;
    ok = (std::fclose(f) == 0) && ok;
    if (ok && std::rename(tmp.c_str(), dst.c_str()) != 0) ok = false;
    if (!ok) (void)std::remove(tmp.c_str());
    return ok;
}

// does <dir>/meta.json list a segment?  (Db::open_or_create would ADD a segment to such a database; this writer only
// creates databases, and replacing the meta would orphan the old segment files - refuse instead)
int existing_segments(const std::string &dir)
{
    std::FILE *f = std::fopen((dir + "/meta.json").c_str(), "rb");
    if (!f) return 0;
    std::string s;
    char buf[4096];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, k);
    std::fclose(f);
    const size_t key = s.find("\"segments\"");
    if (key == std::string::npos) return 0;
    const size_t open = s.find('[', key), close = s.find(']', key);
    if (open == std::string::npos || close == std::string::npos || close < open) return 0;
    return s.find('"', open) < close ? 1 : 0;
}

bool fst_sequential(const EntryVec &e, const std::string &path, std::string *why)
{
    FileSink out(path);
    if (!out.ok()) {
        *why = "cannot create " + path;
        return false;
    }
    FstWriter<FileSink> fst(out, true);
    uint8_t key[17];
    for (size_t i = 0; i < e.size(); i++) {
        e[i].key_bytes(key);
        if (!fst.insert(key, (size_t)e[i].key_len(), i)) {
            *why = "keys not strictly ascending";
            return false;
        }
    }
    fst.finish();
    if (!out.close()) {
        *why = "write failed on " + path;
        return false;
    }
    return true;
}

// the .ids file, sub-tries in parallel (see the top of the file)
bool fst_parallel(const EntryVec &e, const std::string &path, std::string *why)
{
    const size_t n = e.size();
    const bool trace = std::getenv("HB_TRACE_STORE") != nullptr;
    double t_lap = omp_get_wtime();
    auto lap = [&](const char *what) {
        if (trace) std::fprintf(stderr, "[hb store fst] %-24s %8.3f s\n", what, omp_get_wtime() - t_lap);
        t_lap = omp_get_wtime();
    };
    // group = maximal run of keys of length >= 4 with the same first three bytes; shorter keys go through insert()
    struct Group {
        size_t lo, hi;
        bytes sub;
        uint64_t root = 0;
        bool ok = true;
        uint32_t crc = 0; // CRC-32C of `sub`, computed by the thread that built it
    };
    std::vector<Group> groups;
    {
        // boundaries = indices where a group starts, and keys too short to be in one; a group runs from its start to the next boundary.
        // Found by the team, every thread in its own share of the keys (79 M at C4: a third of a second on one thread).
        const int want = n >= (1u << 18) ? hb::host_threads() : 1;
        std::vector<std::vector<uint64_t>> found((size_t)want); // index << 1 | is_start
#pragma omp parallel num_threads(want)
        {
            const size_t nt = (size_t)omp_get_num_threads(), t = (size_t)omp_get_thread_num();
            const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
            std::vector<uint64_t> mine;
            for (size_t i = lo; i < hi; i++) {
                if (e[i].key_len() < 4) {
                    mine.push_back((uint64_t)i << 1);
                    continue;
                }
                if (i == 0 || e[i - 1].key_len() < 4 || (e[i - 1].k0 >> 40) != (e[i].k0 >> 40)) mine.push_back(((uint64_t)i << 1) | 1u);
            }
            if (t < found.size()) found[t].swap(mine);
        }
        std::vector<uint64_t> bnd;
        for (auto &v : found) bnd.insert(bnd.end(), v.begin(), v.end());
        for (size_t k = 0; k < bnd.size(); k++)
            if (bnd[k] & 1u) groups.push_back(Group{(size_t)(bnd[k] >> 1), k + 1 < bnd.size() ? (size_t)(bnd[k + 1] >> 1) : n, {}, 0, true, 0});
    }
    lap("group scan");
#pragma omp parallel for num_threads(hb::host_threads()) schedule(dynamic, 1)
    for (size_t g = 0; g < groups.size(); g++) {
        Group &G = groups[g];
        MemSink mem;
        mem.data.reserve((G.hi - G.lo) * 30 + 64);
        FstWriter<MemSink> sub(mem, false);
        uint8_t key[17];
        for (size_t i = G.lo; i < G.hi && G.ok; i++) {
            e[i].key_bytes(key);
            G.ok = sub.insert(key + 3, (size_t)e[i].key_len() - 3, i - G.lo);
        }
        if (G.ok) G.root = sub.finish_sub();
        G.sub.swap(mem.data);
        G.sub.shrink_to_fit();
        G.crc = crc32c_update(0, G.sub.data(), G.sub.size());
    }
    lap("sub-tries (parallel)");
    FileSink out(path);
    if (!out.ok()) {
        *why = "cannot create " + path;
        return false;
    }
    FstWriter<FileSink> top(out, true);
    uint8_t key[17], last[17];
    size_t g = 0;
    for (size_t i = 0; i < n;) {
        if (g < groups.size() && groups[g].lo == i) {
            Group &G = groups[g++];
            e[i].key_bytes(key);
            e[G.hi - 1].key_bytes(last);
            if (!G.ok || !top.insert_subtrie(key, 3, i, G.sub, G.crc, G.root, G.hi - G.lo, last, (size_t)e[G.hi - 1].key_len())) {
                *why = "keys not strictly ascending";
                return false;
            }
            i = G.hi; // (G.sub stays until out.close() has written it)
        } else {
            e[i].key_bytes(key);
            if (!top.insert(key, (size_t)e[i].key_len(), i)) {
                *why = "keys not strictly ascending";
                return false;
            }
            i++;
        }
    }
    top.finish();
    lap("top levels (sequential)");
    if (!out.close()) {
        *why = "write failed on " + path;
        return false;
    }
    lap("close (parallel pwrite)");
    return true;
}

bool copy_file(const std::string &from, const std::string &to)
{
    std::FILE *a = std::fopen(from.c_str(), "rb"), *b = std::fopen(to.c_str(), "wb");
    bool ok = a && b;
    std::vector<char> buf(1 << 22);
    size_t k;
    while (ok && (k = std::fread(buf.data(), 1, buf.size(), a)) > 0) ok = std::fwrite(buf.data(), 1, k, b) == k;
    if (a) ok = (std::ferror(a) == 0) && ok, std::fclose(a);
    if (b) ok = (std::fclose(b) == 0) && ok;
    return ok;
}

bool pwrite_all(int fd, const uint8_t *p, size_t n, uint64_t off)
{
    while (n) {
        const ssize_t k = ::pwrite(fd, p, n, (off_t)off);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= (size_t)k;
        off += (uint64_t)k;
    }
    return true;
}

// Parallel sort of the entries: one counting-sort pass on the 16 most significant bits in which the keys differ at all
// (NodeIDs are hashes: the class byte is the same for all of them and the two bytes behind it are uniform - 65 536 even
// buckets), then every bucket is sorted on its own, buckets shared out dynamically.  Skewed key sets (tests: small
// integers) only lose balance, never correctness.  (libstdc++'s parallel-mode multiway merge sort took 1.1 s for 5 M
// entries on 8 cores - more than everything else together.)
void parallel_sort(EntryVec &e)
{
    const size_t n = e.size();
    if (n < (1u << 16)) {
        std::sort(e.begin(), e.end());
        return;
    }
    uint64_t diff = 0;
#pragma omp parallel for num_threads(hb::host_threads()) schedule(static) reduction(| : diff)
    for (size_t i = 0; i < n; i++) diff |= e[i].k0 ^ e[0].k0;
    if (!diff) { // all keys share their first 8 bytes: nothing to split on here
        const int before = omp_get_max_threads();
        omp_set_num_threads(hb::host_threads());
        __gnu_parallel::sort(e.begin(), e.end());
        omp_set_num_threads(before); // (the caller's setting is not ours to change)
        return;
    }
    const int top = 63 - __builtin_clzll(diff);     // most significant differing bit of k0
    const int shift = top >= 15 ? top - 15 : 0;      // bucket = 16 bits from there down
    const uint64_t mask = top >= 15 ? 0xFFFFull : ((1ull << (top + 1)) - 1);
    const size_t nb = (size_t)mask + 1;
    const int nt = hb::host_threads();
    std::vector<std::vector<size_t>> hist((size_t)nt, std::vector<size_t>(nb, 0));
    EntryVec tmp(n);
    std::vector<size_t> start(nb + 1, 0);
#pragma omp parallel num_threads(nt)
    {
        // the team may be SMALLER than asked for (a caller's own parallel region around this call, a thread limit): the shares
        // are cut by the team that exists, or entries would be left out
        const int team = omp_get_num_threads();
        const int t = omp_get_thread_num();
        const size_t lo = n * (size_t)t / (size_t)team, hi = n * (size_t)(t + 1) / (size_t)team;
        std::vector<size_t> &h = hist[(size_t)t];
        for (size_t i = lo; i < hi; i++) h[(e[i].k0 >> shift) & mask]++;
#pragma omp barrier
#pragma omp single
        {
            size_t at = 0;
            for (size_t b = 0; b < nb; b++) {
                start[b] = at;
                for (int u = 0; u < team; u++) {
                    const size_t c = hist[(size_t)u][b];
                    hist[(size_t)u][b] = at; // thread u's write position inside bucket b
                    at += c;
                }
            }
            start[nb] = at;
        }
        for (size_t i = lo; i < hi; i++) tmp[h[(e[i].k0 >> shift) & mask]++] = e[i];
#pragma omp barrier
#pragma omp for schedule(dynamic, 16)
        for (size_t b = 0; b < nb; b++)
            if (start[b + 1] - start[b] > 1) std::sort(tmp.begin() + (long)start[b], tmp.begin() + (long)start[b + 1]);
    }
    // bits above `top` are equal in all keys, so bucket order is key order
    e.swap(tmp);
}

// keys of all entries in key-byte order; duplicate ids are an error
int sort_entries(const hb_u128 *ids, uint64_t count, EntryVec *out, char *err, size_t err_len)
{
    if (count >= (1ull << 56)) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: too many entries");
    EntryVec &e = *out;
    e.resize(count);
#pragma omp parallel for num_threads(hb::host_threads()) schedule(static)
    for (uint64_t i = 0; i < count; i++) {
        uint8_t key[17] = {0};
        (void)varint_u128(((unsigned __int128)ids[i].hi << 64) | ids[i].lo, key);
        uint64_t k0 = 0, k1 = 0;
        for (int b = 0; b < 8; b++) k0 = (k0 << 8) | key[b];
        for (int b = 0; b < 8; b++) k1 = (k1 << 8) | key[8 + b];
        e[i] = Entry{k0, k1, ((uint64_t)key[16] << 56) | i};
    }
    // keys in ascending byte order of their encodings (LiveSegment is a BTreeMap<Vec<u8>, _>, lib.rs:108-110)
    parallel_sort(e);
    bool dup = false;
#pragma omp parallel for num_threads(hb::host_threads()) schedule(static) reduction(|| : dup)
    for (uint64_t i = 1; i < count; i++) dup = dup || e[i].same_key(e[i - 1]);
    if (dup) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: duplicate NodeID");
    return HB_OK;
}

// .blobs and .bid of one database from the sorted entries (SegmentWriter::insert, segment.rs:66-75)
bool write_blobs(const EntryVec &e, const void *values, int value_kind, const std::string &base, std::string *why)
{
    const uint64_t count = e.size();
    const uint64_t block = 1ull << 16;
    const uint64_t nblocks = (count + block - 1) / block;
    std::vector<uint64_t> block_bytes(nblocks + 1, 0);
    auto entry_len = [&](const Entry &x) -> uint64_t {
        uint64_t v = 8;
        if (value_kind == HB_STORE_U64) {
            uint8_t tmp[17];
            v = varint_u128(((const uint64_t *)values)[x.index()], tmp);
        }
        return (uint64_t)x.key_len() + v;
    };
#pragma omp parallel for num_threads(hb::host_threads()) schedule(static)
    for (uint64_t b = 0; b < nblocks; b++) {
        uint64_t s = 0;
        for (uint64_t i = b * block; i < std::min(count, (b + 1) * block); i++) s += entry_len(e[i]);
        block_bytes[b + 1] = s;
    }
    for (uint64_t b = 0; b < nblocks; b++) block_bytes[b + 1] += block_bytes[b];
    const std::string pb = base + ".blobs", pi = base + ".bid";
    const int fb = ::open(pb.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666), fi = ::open(pi.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    bool ok = fb >= 0 && fi >= 0;
    if (!ok) *why = "cannot create " + (fb < 0 ? pb : pi);
    std::atomic<bool> io_ok{true};
    if (ok) {
#pragma omp parallel num_threads(hb::host_threads())
        {
            bytes blob, bid;
#pragma omp for schedule(dynamic, 4)
            for (uint64_t b = 0; b < nblocks; b++) {
                blob.clear();
                bid.clear();
                uint64_t offset = block_bytes[b];
                const uint64_t lo = b * block, hi = std::min(count, (b + 1) * block);
                bid.resize((hi - lo) * 32);
                uint8_t key[17], val[17];
                for (uint64_t i = lo; i < hi; i++) {
                    const Entry &x = e[i];
                    x.key_bytes(key);
                    const uint64_t kl = (uint64_t)x.key_len();
                    uint64_t vl = 8;
                    if (value_kind == HB_STORE_F64) std::memcpy(val, &((const double *)values)[x.index()], 8); // little-endian host
                    else vl = varint_u128(((const uint64_t *)values)[x.index()], val);
                    blob.insert(blob.end(), key, key + kl);
                    blob.insert(blob.end(), val, val + vl);
                    const uint64_t ptr[4] = {offset, offset + kl, offset + kl, offset + kl + vl}; // BlobPointer: key range, value range
                    std::memcpy(bid.data() + (i - lo) * 32, ptr, 32);
                    offset += kl + vl;
                }
                if (!pwrite_all(fb, blob.data(), blob.size(), block_bytes[b]) || !pwrite_all(fi, bid.data(), bid.size(), lo * 32)) io_ok = false;
            }
        }
        if (!io_ok) {
            ok = false;
            *why = "write failed on " + pb + " / " + pi;
        }
    }
    if (fb >= 0 && ::close(fb) != 0 && ok) ok = false, *why = "write failed on " + pb;
    if (fi >= 0 && ::close(fi) != 0 && ok) ok = false, *why = "write failed on " + pi;
    return ok;
}

bool write_file(const std::string &path, const bytes &b)
{
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = b.empty() || std::fwrite(b.data(), 1, b.size(), f) == b.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

struct Target {
    std::string dir;
    const void *values;
    int kind;
};

// one key set, any number of databases over it
int write_dbs(const std::vector<Target> &targets, const hb_u128 *ids, uint64_t count, char *err, size_t err_len, EntryVec *presorted = nullptr)
{
    if (count && !ids && !presorted) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: NULL array with count > 0");
    for (const Target &t : targets) {
        if (t.dir.empty()) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: dir is empty");
        if (count && !t.values) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: NULL array with count > 0");
        if (t.kind != HB_STORE_F64 && t.kind != HB_STORE_U64) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: unknown value kind");
        if (!make_dirs(t.dir)) return fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot create directory " + t.dir);
        if (existing_segments(t.dir))
            return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: " + t.dir + "/meta.json already lists segments; this writer creates databases, it does not "
                                                          "append to them (Db::open_or_create would add a segment): choose an empty directory");
    }
    if (count == 0) { // Db::commit with an empty live segment writes nothing (lib.rs:376-379); open_or_create saved an empty Meta
        for (const Target &t : targets)
            if (!write_meta(t.dir, "{\n  \"segments\": []\n}")) return fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot write " + t.dir + "/meta.json");
        return HB_OK;
    }
    const bool trace = std::getenv("HB_TRACE_STORE") != nullptr; // phase times on stderr
    double t_lap = omp_get_wtime();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const double t = omp_get_wtime();
        std::fprintf(stderr, "[hb store] %-28s %8.3f s  (%d threads)\n", what, t - t_lap, hb::host_threads());
        t_lap = t;
    };
    EntryVec entries;
    if (presorted) {
        // the key order was computed elsewhere (hb_store_harmonic_results: a radix sort on the device, hb_ingest.hip gpu_store_keys):
        // trusted for order only after the same check the host sort ends with - strictly ascending keys, every index in range
        entries.swap(*presorted);
        if (entries.size() != count) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: presorted key set of the wrong size");
        bool bad = false;
#pragma omp parallel for num_threads(hb::host_threads()) schedule(static) reduction(|| : bad)
        for (uint64_t i = 0; i < count; i++) bad = bad || entries[i].index() >= count || (i && !(entries[i - 1] < entries[i])) || (i && entries[i].same_key(entries[i - 1]));
        if (bad) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: duplicate NodeID (or a presorted key set that is not ascending)");
        lap("keys (presorted) + check");
    } else {
        int rc = sort_entries(ids, count, &entries, err, err_len);
        if (rc != HB_OK) return rc;
        lap("keys + sort");
    }
    // bloom filter: SegmentWriter::new(num_items, ..): BytesBloomFilter::new(num_items, 0.01), segment.rs:56-59
    bytes blm;
    {
        Bloom bloom(count);
        const uint64_t share = 1ull << 16;
        const uint64_t nshares = (count + share - 1) / share;
#pragma omp parallel for num_threads(hb::host_threads()) schedule(dynamic, 1)
        for (uint64_t sidx = 0; sidx < nshares; sidx++)
            bloom.insert_range(sidx * share, std::min(count, (sidx + 1) * share), [&](uint64_t i, uint8_t *key) -> size_t {
                entries[i].key_bytes(key);
                return (size_t)entries[i].key_len();
            });
        bloom.serialize(blm);
    }
    lap("bloom");
    const char *mode = std::getenv("HB_STORE_FST");
    const bool sequential = mode && std::strcmp(mode, "sequential") == 0;
    // A call that fails leaves nothing behind: every file it wrote is removed again, and the meta.json files - what makes a
    // directory a database, and what makes this writer refuse it next time - are written only after the segment files of ALL targets
    // are complete (hb_store_harmonic: a failure in `harmonic_rank` must not leave a `harmonic` that blocks the retry).
    struct Undo {
        std::vector<std::string> paths;
        bool armed = true;
        ~Undo()
        {
            if (armed)
                for (const std::string &f : paths) (void)::unlink(f.c_str());
        }
    } undo;
    std::vector<std::string> uuids, bases;
    for (const Target &t : targets) {
        uuids.push_back(uuid_v4());
        bases.push_back(t.dir + "/" + uuids.back());
        for (const char *ext : {".blobs", ".bid", ".ids", ".blm"}) undo.paths.push_back(bases.back() + ext);
    }
    // [r6] the fst is built BESIDE the blob files: `.blobs` / `.bid` are page-cache copies behind one inode lock per file (two writers at a
    // time whatever the team size), the fst is CPU work on the same keys - one after the other they were 0.7 + 3.5 + 1.2..1.8 s at C4
    // (profiles/r06l_store_phases_C4_then_C3.txt).  A second thread (with an OpenMP team of its own) builds `.ids` of the first target
    // while this one writes the blob files of all targets; joined before anything is undone or declared complete.
    struct FstJob {
        bool ok = true;
        int code = HB_ERR_IO;
        std::string why;
        double seconds = 0.0;
    } job;
    std::thread fst_thread;
    struct Joiner {
        std::thread &t;
        ~Joiner()
        {
            if (t.joinable()) t.join();
        }
    } joiner{fst_thread}; // (declared after `undo`: it joins before the undo list is walked)
    if (!targets.empty()) {
        fst_thread = std::thread([&]() {
            const double t0 = omp_get_wtime();
            try {
                const std::string path = bases[0] + ".ids";
                if (!(sequential ? fst_sequential(entries, path, &job.why) : fst_parallel(entries, path, &job.why))) {
                    job.ok = false;
                    job.code = job.why.find("ascending") != std::string::npos ? HB_ERR_INVALID : HB_ERR_IO;
                }
            } catch (const std::bad_alloc &) {
                job.ok = false;
                job.code = HB_ERR_NOMEM;
                job.why = "out of host memory";
            } catch (const std::exception &e) {
                job.ok = false;
                job.code = HB_ERR_INVALID;
                job.why = e.what();
            }
            job.seconds = omp_get_wtime() - t0;
        });
    }
    for (size_t k = 0; k < targets.size(); k++) {
        std::string why;
        if (!write_blobs(entries, targets[k].values, targets[k].kind, bases[k], &why)) return fail(err, err_len, HB_ERR_IO, "hb_store_write: " + why);
        lap(".blobs + .bid");
    }
    if (fst_thread.joinable()) fst_thread.join();
    if (!job.ok) return fail(err, err_len, job.code, "hb_store_write: " + job.why);
    if (trace) std::fprintf(stderr, "[hb store] %-28s %8.3f s  (beside the blob files; %d threads)\n", ".ids (fst)", job.seconds, hb::host_threads());
    lap(".ids (fst): waited for");
    for (size_t k = 0; k < targets.size(); k++) {
        if (k && ::link((bases[0] + ".ids").c_str(), (bases[k] + ".ids").c_str()) != 0 && !copy_file(bases[0] + ".ids", bases[k] + ".ids")) {
            // same keys, same values 0 .. count-1: the same map - a second NAME for the same bytes where the file system allows it (both
            // databases live under one output directory; segment files are written once and only ever read, replaced as a whole or
            // removed: blob_id_index.rs:43-60 maps them read-only), a copy otherwise (C4: 1.3 GB, 0.4 s)
            return fail(err, err_len, HB_ERR_IO, "hb_store_write: write failed on " + bases[k] + ".ids");
        }
        if (!write_file(bases[k] + ".blm", blm)) return fail(err, err_len, HB_ERR_IO, "hb_store_write: write failed on " + bases[k] + ".blm");
    }
    lap(".ids link + .blm");
    // Meta { segments: [uuid] } through serde_json::to_string_pretty (lib.rs:292-297); last, so that a database whose
    // meta.json exists is complete
    for (size_t k = 0; k < targets.size(); k++) {
        undo.paths.push_back(targets[k].dir + "/meta.json"); // (it replaces at most an empty `{"segments": []}`: removing it again loses nothing)
        if (!write_meta(targets[k].dir, "{\n  \"segments\": [\n    \"" + uuids[k] + "\"\n  ]\n}"))
            return fail(err, err_len, HB_ERR_IO, "hb_store_write: cannot write " + targets[k].dir + "/meta.json");
    }
    undo.armed = false;
    return HB_OK;
}

template <class F>
int guarded(char *err, size_t err_len, F &&f)
{
    if (err && err_len) err[0] = 0;
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(err, err_len, HB_ERR_NOMEM, "hb_store_write: out of host memory");
    } catch (const std::exception &e) {
        return fail(err, err_len, HB_ERR_INVALID, std::string("hb_store_write: ") + e.what());
    }
}

} // namespace

// hb_internal.h: store_harmonic with the key order already computed (StoreKey = Entry, three words)
namespace hb {
int store_harmonic_presorted(const char *output, StoreKeyVec *sorted, const double *centralities, const uint64_t *ranks, char *err, size_t err_len)
{
    return guarded(err, err_len, [&]() -> int {
        if (!output || !*output || !sorted) return fail(err, err_len, HB_ERR_INVALID, "hb_store_harmonic: output is empty");
        const std::string out(output);
        const uint64_t count = sorted->size();
        return write_dbs({Target{out + "/harmonic", centralities, HB_STORE_F64}, Target{out + "/harmonic_rank", ranks, HB_STORE_U64}}, nullptr, count, err, err_len, sorted);
    });
}
} // namespace hb

extern "C" int hb_store_write(const char *dir, const hb_u128 *ids, const void *values, int value_kind, uint64_t count, char *err, size_t err_len)
{
    return guarded(err, err_len, [&]() -> int {
        if (!dir || !*dir) return fail(err, err_len, HB_ERR_INVALID, "hb_store_write: dir is empty");
        return write_dbs({Target{dir, values, value_kind}}, ids, count, err, err_len);
    });
}

extern "C" int hb_store_harmonic(const char *output, const hb_u128 *ids, const double *centralities, const uint64_t *ranks, uint64_t count,
                                 char *err, size_t err_len)
{
    return guarded(err, err_len, [&]() -> int {
        if (!output || !*output) return fail(err, err_len, HB_ERR_INVALID, "hb_store_harmonic: output is empty");
        const std::string out(output);
        return write_dbs({Target{out + "/harmonic", centralities, HB_STORE_F64}, Target{out + "/harmonic_rank", ranks, HB_STORE_U64}}, ids, count, err, err_len);
    });
}
