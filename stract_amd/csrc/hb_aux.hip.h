// hb_aux.hip.h - helpers, the changed-only exchanges, the reference-tail mode, normalisation and state checksums.
// Part of the device code of stract_amd/csrc/hb_kernels.hip.h (included from there).
#pragma once

namespace hbk {

// ---- helpers ---------------------------------------------------------------------------
// Structural check of a finished work layout, run once per load on every build (one streaming pass over the row pointers and
// the source lists: 0.3 ms at C3, 3 ms at C4): every index a pass kernel will ever gather through is in range BEFORE the first
// pass runs, so a planner or ingest defect is an error code at load time, not a GPU memory fault three kernels later.
// bad[0] row pointers not monotone / beyond src_len, [1] source id >= rows_total (kNone inside a row included),
// [2] sources of one row of mixed kind (node and virtual ids), [3] a virtual row reading itself or a later virtual row
__global__ __launch_bounds__(256) void validate_plan_kernel(const uint64_t *row_ptr, const uint32_t *src, uint64_t rows_total, uint64_t n_pad,
                                                            uint64_t src_len, unsigned long long *bad)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nq = (uint64_t)gridDim.x * 64; // quads in the grid
    const int q = threadIdx.x & 3;
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (uint64_t row = t >> 2; row < rows_total; row += nq) {
        const uint64_t b = row_ptr[row], e = row_ptr[row + 1];
        if (b > e || e > src_len) {
            b0 += (q == 0);
            continue;
        }
        if (row + 1 == rows_total && e != src_len) b0 += (q == 0);
        const bool first_virtual = b < e && src[b] >= n_pad;
        for (uint64_t k = b + q; k < e; k += 4) {
            const uint32_t s = src[k];
            if (s >= rows_total) b1++;
            else {
                if ((s >= n_pad) != first_virtual) b2++;
                if (s >= n_pad && (row < n_pad ? false : s >= row)) b3++; // partials are produced level by level, bottom-up
            }
        }
    }
    if (b0) atomicAdd(&bad[0], (unsigned long long)b0);
    if (b1) atomicAdd(&bad[1], (unsigned long long)b1);
    if (b2) atomicAdd(&bad[2], (unsigned long long)b2);
    if (b3) atomicAdd(&bad[3], (unsigned long long)b3);
}

__global__ __launch_bounds__(256) void hll_size_kernel(const uint4 *regs, uint64_t count, uint64_t *out,
                                                       const double *raw, const double *bias, const uint8_t *lc)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = t >> 2;
    const uint64_t rows_pad = (count + 15) & ~15ull;
    if (row >= rows_pad) return;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < count) v = regs[t];
    const uint64_t sz = hll_size_quad(v, raw, bias, lc);
    if (row < count && (t & 3) == 0) out[row] = sz;
}

// wr = max(wr, other) byte-wise: all-reduce(max) between logical ranks on one device
__global__ __launch_bounds__(256) void merge_max_kernel(uint4 *dst, const uint4 *other, uint64_t count4)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (uint64_t)gridDim.x * 256) {
        uint4 a = dst[i];
        const uint4 b = other[i];
        Acc acc;
        acc_zero(acc);
        acc_merge(acc, a);
        acc_merge(acc, b);
        dst[i] = acc_value(acc);
    }
}

// dst |= src word-wise (union of the ranks' locally-changed bitmaps)
__global__ __launch_bounds__(256) void or_words_kernel(uint32_t *dst, const uint32_t *src, uint64_t words)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (uint64_t)gridDim.x * 256) dst[i] |= src[i];
}
// edge partition, changed-only: the all-reduced packed rows go back to their places (quad per row of [0, n_pad))
__global__ __launch_bounds__(256) void unpack_rows_kernel(uint4 *wr, const uint32_t *bits, const uint64_t *prefix, uint64_t n_pad, const uint4 *pack)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = t >> 2;
    if (row >= n_pad) return;
    const uint32_t w = bits[row >> 5], b = (uint32_t)(row & 31u);
    if (!((w >> b) & 1u)) return;
    const uint64_t pos = prefix[row >> 5] + (uint64_t)__popc(w & ((1u << b) - 1u));
    wr[row * 4 + (t & 3)] = pack[pos * 4 + (t & 3)];
}

// ---- changed-only exchange (destination partition, HB_FLAG_CHANGED_ONLY) --------------------------------------
// After the changed bits of all slices are known everywhere, only the counters that changed travel: every rank
// packs the changed rows of its slice (ascending row order; position = rank of the row's bit among all set bits,
// from a prefix sum over the bitmap words), the packed runs are broadcast, and the receivers scatter them.
__global__ __launch_bounds__(256) void popcount_words_kernel(const uint32_t *bits, uint64_t words, uint32_t *out)
{
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (uint64_t)gridDim.x * 256) out[w] = __popc(bits[w]);
}
// [r5] rank_cnt[k] = (k == rank) ? rows this rank's fused launch changed (word 0 of its counter stripes) : 0.  Summed over the ranks next
// to the pass counters, the host gets every rank's run length with the ONE read-back a pass has anyway - the packed runs'
// offsets no longer cost a round trip of their own.
__global__ __launch_bounds__(64) void rank_count_kernel(const unsigned long long *counters, unsigned long long *rank_cnt, int world, int rank)
{
    unsigned long long v = threadIdx.x < kStripes ? counters[4 * threadIdx.x] : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    v = __shfl(v, 0);
    for (int k = threadIdx.x; k < world; k += 64) rank_cnt[k] = k == rank ? v : 0ull;
}
// ---- 6 bits per register on the wire [r6] ------------------------------------------------------------------------------------
// A counter of this path only ever holds registers from {0} u [1, 58] u {65}: HyperLogLog::add sets p = lzcnt64(w) + 1 with
// w = hash << 6 (hyperloglog.rs:4385-4396) - w's six low bits are zero, so a non-zero w is >= 2^6 and lzcnt <= 57 (p <= 58), and
// w == 0 gives lzcnt 64 (p = 65); merges are register-wise maxima of such values (:4531-4535), which stay in the set.  The values
// 59..64 cannot occur, so 6 bits hold a register EXACTLY: code = r for r <= 58, code 63 for r = 65.  A packed counter is 48 bytes
// instead of 64: 25 % fewer bytes in every packed exchange of the destination partition, nothing approximated, no escape list.
// (A register outside the set - impossible through the C ABI's graph input - is caught by the index-check builds.)
// lane q of a row's quad holds registers 16 q .. 16 q + 15 as a uint4 and produces / consumes bytes 12 q .. 12 q + 11 of the 48.
struct Packed12 {
    uint32_t w[3];
};
__device__ __forceinline__ Packed12 pack6_quarter(const uint4 &v)
{
    const uint32_t in[4] = {v.x, v.y, v.z, v.w};
    uint64_t lo = 0; // codes 0..9 (60 bits) + the low 4 bits of code 10
    uint32_t hi = 0; // the rest: 2 + 5 * 6 = 32 bits
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t r = (in[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        HB_DBG_ASSERT(r <= 58u || r == 65u);
        const uint64_t code = r == 65u ? 63u : (r & 63u);
        if (i < 10) lo |= code << (6 * i);
        else if (i == 10) {
            lo |= (code & 15u) << 60;
            hi |= (uint32_t)(code >> 4);
        } else hi |= (uint32_t)code << (2 + 6 * (i - 11));
    }
    Packed12 p;
    p.w[0] = (uint32_t)lo;
    p.w[1] = (uint32_t)(lo >> 32);
    p.w[2] = hi;
    return p;
}
__device__ __forceinline__ uint4 unpack6_quarter(const Packed12 &p)
{
    const uint64_t lo = ((uint64_t)p.w[1] << 32) | p.w[0];
    const uint32_t hi = p.w[2];
    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t code;
        if (i < 10) code = (uint32_t)(lo >> (6 * i)) & 63u;
        else if (i == 10) code = ((uint32_t)(lo >> 60) & 15u) | ((hi & 3u) << 4);
        else code = (hi >> (2 + 6 * (i - 11))) & 63u;
        const uint32_t r = code == 63u ? 65u : code;
        out[i >> 2] |= r << (8 * (i & 3));
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}
// the packed forms of pack_changed_kernel / unpack_changed_kernel below: 48-byte rows at pack6[pos * 48]
__global__ __launch_bounds__(256) void pack6_changed_kernel(const uint4 *wr, const uint32_t *bits, const uint64_t *prefix, uint64_t row_lo, uint64_t row_hi,
                                                            uint32_t *pack6)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = row_lo + (t >> 2);
    if (row >= row_hi) return;
    const uint32_t w = bits[row >> 5], b = (uint32_t)(row & 31u);
    if (!((w >> b) & 1u)) return;
    const uint64_t pos = prefix[row >> 5] + (uint64_t)__popc(w & ((1u << b) - 1u));
    const Packed12 p = pack6_quarter(wr[row * 4 + (t & 3)]);
    uint32_t *dst = pack6 + pos * 12 + (t & 3) * 3;
    dst[0] = p.w[0];
    dst[1] = p.w[1];
    dst[2] = p.w[2];
}
__global__ __launch_bounds__(256) void unpack6_changed_kernel(uint4 *wr, const uint4 *rd, const uint32_t *bits_now, const uint32_t *bits_prev, const uint64_t *prefix,
                                                              uint64_t row_lo, uint64_t row_hi, const uint32_t *pack6)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = row_lo + (t >> 2);
    if (row >= row_hi) return;
    const uint32_t w = bits_now[row >> 5], b = (uint32_t)(row & 31u);
    if ((w >> b) & 1u) {
        const uint64_t pos = prefix[row >> 5] + (uint64_t)__popc(w & ((1u << b) - 1u));
        const uint32_t *src = pack6 + pos * 12 + (t & 3) * 3;
        Packed12 p;
        p.w[0] = src[0];
        p.w[1] = src[1];
        p.w[2] = src[2];
        wr[row * 4 + (t & 3)] = unpack6_quarter(p);
    } else if ((bits_prev[row >> 5] >> b) & 1u) {
        wr[row * 4 + (t & 3)] = rd[row * 4 + (t & 3)];
    }
}

// quad per row of [row_lo, row_hi)
__global__ __launch_bounds__(256) void pack_changed_kernel(const uint4 *wr, const uint32_t *bits, const uint64_t *prefix, uint64_t row_lo,
                                                           uint64_t row_hi, uint4 *pack)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = row_lo + (t >> 2);
    if (row >= row_hi) return;
    const uint32_t w = bits[row >> 5], b = (uint32_t)(row & 31u);
    if (!((w >> b) & 1u)) return;
    const uint64_t pos = prefix[row >> 5] + (uint64_t)__popc(w & ((1u << b) - 1u));
    pack[pos * 4 + (t & 3)] = wr[row * 4 + (t & 3)];
}
// foreign rows [row_lo, row_hi): changed now -> take the packed counter; changed in the previous pass only -> the
// other buffer is two passes old, carry the current value over (lazy double buffer, see pass_kernel)
__global__ __launch_bounds__(256) void unpack_changed_kernel(uint4 *wr, const uint4 *rd, const uint32_t *bits_now, const uint32_t *bits_prev,
                                                             const uint64_t *prefix, uint64_t row_lo, uint64_t row_hi, const uint4 *pack)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = row_lo + (t >> 2);
    if (row >= row_hi) return;
    const uint32_t w = bits_now[row >> 5], b = (uint32_t)(row & 31u);
    if ((w >> b) & 1u) {
        const uint64_t pos = prefix[row >> 5] + (uint64_t)__popc(w & ((1u << b) - 1u));
        wr[row * 4 + (t & 3)] = pack[pos * 4 + (t & 3)];
    } else if ((bits_prev[row >> 5] >> b) & 1u) {
        wr[row * 4 + (t & 3)] = rd[row * 4 + (t & 3)];
    }
}

// ---- reference-tail mode (HB_FLAG_REFERENCE_TAIL): the changed-node machinery of the reference as written -----------
// U64BloomFilter::insert_u128 (bloom/src/lib.rs:85-98): slot = (low 64 bits of the id * LARGE_PRIME) % num_bits.
constexpr unsigned long long kBloomPrime = 11400714819323198549ull;
__device__ __forceinline__ uint64_t bloom_slot(uint64_t id_low, uint64_t num_bits) { return (id_low * kBloomPrime) % num_bits; }

// new_changed_nodes of one pass: a bit per slot of every changed node (harmonic.rs:145,103)
__global__ __launch_bounds__(256) void bloom_insert_kernel(const uint32_t *bits, const uint64_t *id_low, uint64_t n_pad, uint64_t num_bits,
                                                           uint32_t *bloom)
{
    for (uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; row < n_pad; row += (uint64_t)gridDim.x * 256) {
        if (!((bits[row >> 5] >> (row & 31u)) & 1u)) continue;
        const uint64_t s = bloom_slot(id_low[row], num_bits);
        atomicOr(&bloom[s >> 5], 1u << (s & 31u));
    }
}
// bit_vec.count_ones() (bloom/src/lib.rs:109)
__global__ __launch_bounds__(256) void bloom_count_kernel(const uint32_t *bloom, uint64_t words, unsigned long long *out)
{
    unsigned long long c = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (uint64_t)gridDim.x * 256) c += __popc(bloom[w]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// changed_nodes.contains_u128(edge.from) (harmonic.rs:133) for every node: the frontier WITH the filter's false
// positives - they are results-inert only as long as no tail pass has skipped host-level edges (hb_api.hip)
__global__ __launch_bounds__(256) void bloom_frontier_kernel(const uint32_t *bloom, const uint64_t *id_low, const uint32_t *sid_of,
                                                             uint64_t n_pad, uint64_t num_bits, uint32_t *bits)
{
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; // n_pad is a multiple of 64: whole waves
    if (row >= n_pad) return;
    bool in = false;
    if (sid_of[row] != kNone) {
        const uint64_t s = bloom_slot(id_low[row], num_bits);
        in = (bloom[s >> 5] >> (s & 31u)) & 1u;
    }
    const uint64_t bal = __ballot(in);
    if ((threadIdx.x & 63) == 0) {
        bits[row >> 5] = (uint32_t)bal;
        bits[(row >> 5) + 1] = (uint32_t)(bal >> 32);
    }
}
// exact_changed_nodes (harmonic.rs:146-148,105) as a list of device rows; order is irrelevant (max is commutative)
__global__ __launch_bounds__(256) void changed_list_kernel(const uint32_t *bits, uint64_t n_pad, uint32_t *list, unsigned int *count, uint32_t cap)
{
    for (uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; row < n_pad; row += (uint64_t)gridDim.x * 256) {
        if (!((bits[row >> 5] >> (row & 31u)) & 1u)) continue;
        const unsigned int k = atomicAdd(count, 1u);
        if (k < cap) list[k] = (uint32_t)row;
    }
}
__device__ __forceinline__ uint32_t bytes_max(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const uint32_t x = (a >> k) & 0xFFu, y = (b >> k) & 0xFFu;
        r |= (x > y ? x : y) << k;
    }
    return r;
}
// update_changed_counters (harmonic.rs:75-114): for every changed node u and every record (u -> v) the forward-links
// query returns: counters.new[v] = max(counters.new[v], counters.old[u]) register-wise.  One wave per changed node,
// 4 records x 16 words at a time; targets are shared between nodes, hence the compare-and-swap.  wr = copy of rd.
__global__ __launch_bounds__(256) void tail_merge_kernel(const uint32_t *list, const unsigned int *count, const uint64_t *tail_ptr,
                                                         const uint32_t *tail_to, const uint32_t *rd, uint32_t *wr)
{
    const uint32_t lane = threadIdx.x & 63u, k = lane >> 4, w = lane & 15u;
    const uint32_t total = *count;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < total; i += gridDim.x * 4) {
        const uint64_t u = list[i];
        const uint32_t from = rd[u * 16 + w];
        const uint64_t e = tail_ptr[u + 1];
        for (uint64_t j = tail_ptr[u] + k; j < e; j += 4) {
            uint32_t *dst = &wr[(uint64_t)tail_to[j] * 16 + w];
            uint32_t old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                const uint32_t nw = bytes_max(old, from);
                if (nw == old) break;
                const uint32_t prev = atomicCAS(dst, old, nw);
                if (prev == old) break;
                old = prev;
            }
        }
    }
}

// ---- normalize_centralities (harmonic.rs:178-195) -----------------------------------------
// f64::from(KahanSum) = sum (kahan_sum.rs:35-39); kept iff > 0.0, then / norm, non-finite -> 0.0; absent nodes are marked -1.0.
__device__ __forceinline__ double normalized_centrality(double s, double norm, bool &kept)
{
    double v = -1.0;
    kept = s > 0.0;
    if (kept) {
        v = s / norm;
        if (!(fabs(v) <= 1.7976931348623157e308)) v = 0.0; // is_finite
    }
    return v;
}
// out[sid] for sid in ascending-NodeID order
__global__ __launch_bounds__(256) void finish_kernel(const double *ksum, const uint32_t *dev_of, uint64_t n,
                                                     double norm, double *out, unsigned long long *count)
{
    unsigned long long kept = 0;
    for (uint64_t sid = (uint64_t)blockIdx.x * 256 + threadIdx.x; sid < n; sid += (uint64_t)gridDim.x * 256) {
        bool k;
        out[sid] = normalized_centrality(ksum[dev_of[sid]], norm, k);
        kept += k;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    const unsigned long long v[4] = {kept, 0, 0, 0};
    block_add_counters(count, v, 0x1u); // striped: the host sums word 0 of every stripe
}

// ---- the compact result image [r6] ----------------------------------------------------------------------------------------------
// A node without in-edges never changes its counter, so its centrality stays 0 and it is never a result (harmonic.rs:178-195 keeps
// centrality > 0): on a single rank the host-bound image therefore holds one f64 per node WITH in-edges only, in ascending-NodeID order
// ("cid" = rank of the node among those) - at C4 79.0 M of 99.2 M entries: 632 instead of 794 MB over the link for the snapshot whose
// download hb_finish waits for, and as many fewer scattered stores in the snapshot kernel.  Built once per load: flags in NodeID order
// -> exclusive scan -> cid per device row (kNone = no in-edges: the row is skipped by results_sync_kernel); the flags also go to the
// host as a bitmap, from which hb_result_copy / hb_result_top find the NodeID of every compact entry.
__global__ __launch_bounds__(256) void in_flags_kernel(const uint64_t *row_ptr, const uint32_t *sid_of, uint64_t n_pad, uint32_t *flags)
{
    for (uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; row < n_pad; row += (uint64_t)gridDim.x * 256) {
        const uint32_t sd = sid_of[row];
        if (sd != kNone) flags[sd] = row_ptr[row + 1] > row_ptr[row] ? 1u : 0u;
    }
}
__global__ __launch_bounds__(256) void cid_of_kernel(const uint64_t *row_ptr, const uint32_t *sid_of, const uint64_t *cpos, uint64_t n_pad, uint32_t *cid_of)
{
    for (uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; row < n_pad; row += (uint64_t)gridDim.x * 256) {
        const uint32_t sd = sid_of[row];
        cid_of[row] = (sd != kNone && row_ptr[row + 1] > row_ptr[row]) ? (uint32_t)cpos[sd] : kNone;
    }
}
// words[w] bit b = flags[64 w + b] (the wave's ballot IS the word); `words` covers ceil(n / 64) entries
__global__ __launch_bounds__(256) void pack_flags_kernel(const uint32_t *flags, uint64_t n, unsigned long long *words)
{
    const uint64_t nw = (n + 63) >> 6;
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (uint64_t)gridDim.x * 4) { // wave-uniform trip count
        const uint64_t i = (w << 6) + (threadIdx.x & 63);
        const unsigned long long m = __ballot(i < n && flags[i] != 0u);
        if ((threadIdx.x & 63) == 0) words[w] = m;
    }
}

// ---- results that travel while the passes still run (hb_api.hip: results_stage / hb_finish) --------------------------------
// The host needs one f64 per node at the end (HarmonicCentrality's map); a download of n x 8 bytes AFTER the last pass is
// 7-8 % of a whole run at BASELINE sizes (C4: 794 MB = 14 ms over the link against 212 ms).  Most nodes' sums are final long
// before the loop ends (a node's sum stops moving one pass after its counter does), so the bulk is shipped on a second stream
// while the next pass runs, and what still moved afterwards follows as a short list.  `sent[row]` = the sum (device order)
// the host-bound image `out` (ascending-NodeID order, normalised) was last built from.  One streaming pass in DEVICE order:
// rows whose sum differs bitwise from sent[] (all rows when `all`) refresh out[sid] and sent[], and - if a list is wanted - are
// appended as (sid, value); a block reserves list space once per 2048 rows (same-address atomics sustain ~90/us).
// counts[0] = entries appended (may exceed cap: the list is then incomplete and the caller ships `out` whole).
// kept (word 0 of the striped `count`) = nodes with centrality > 0 among ALL rows = the result count.
constexpr int kSyncPerThread = 8;
__global__ __launch_bounds__(256) void results_sync_kernel(const double *ksum, double *sent, const uint32_t *sid_of, uint64_t n_pad, double norm,
                                                           int all, double *out, uint32_t *list_sid, double *list_val, unsigned long long cap,
                                                           unsigned long long *counts, unsigned long long *count)
{
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long kept = 0;
    const uint64_t span = 256ull * kSyncPerThread;
    for (uint64_t base = (uint64_t)blockIdx.x * span; base < n_pad; base += (uint64_t)gridDim.x * span) { // block-uniform trip count
        uint32_t mask = 0, sid[kSyncPerThread];
        double val[kSyncPerThread];
#pragma unroll
        for (int j = 0; j < kSyncPerThread; j++) {
            const uint64_t row = base + (uint64_t)j * 256 + threadIdx.x;
            sid[j] = kNone;
            val[j] = 0.0;
            if (row < n_pad) {
                const uint32_t sd = sid_of[row];
                if (sd != kNone) {
                    const double s = ksum[row];
                    bool k;
                    const double v = normalized_centrality(s, norm, k);
                    kept += k;
                    if (all || __double_as_longlong(s) != __double_as_longlong(sent[row])) {
                        sent[row] = s;
                        out[sd] = v;
                        mask |= 1u << j;
                        sid[j] = sd;
                        val[j] = v;
                    }
                }
            }
        }
        if (list_sid) {
            const uint32_t mine = __popc(mask);
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t a = __shfl_up(incl, off);
                if (lane >= off) incl += a;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
                s_base = tot ? atomicAdd(counts, (unsigned long long)tot) : 0ull;
            }
            __syncthreads();
            unsigned long long pos = s_base + (incl - mine);
            for (int w = 0; w < wave; w++) pos += s_wave[w];
#pragma unroll
            for (int j = 0; j < kSyncPerThread; j++) {
                if ((mask >> j) & 1u) {
                    if (pos < cap) {
                        list_sid[pos] = sid[j];
                        list_val[pos] = val[j];
                    }
                    pos++;
                }
            }
            __syncthreads(); // s_wave / s_base are rewritten by the next iteration
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    const unsigned long long v[4] = {kept, 0, 0, 0};
    block_add_counters(count, v, 0x1u);
}

// Order-independent checksums of the state (hb_debug_state_hash; same function as
// oracle/hb_oracle.c hbo_dense_state_hash): node sid contributes mixes of (sid, its 8 register
// words) and of (sid, sum bits, err bits); contributions are added mod 2^64.
__device__ __forceinline__ uint64_t hash_mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
__global__ __launch_bounds__(256) void state_hash_kernel(const uint4 *regs, const double *ksum, const double *kerr,
                                                         const uint32_t *dev_of, uint64_t n, unsigned long long *out)
{
    unsigned long long hr = 0, hk = 0;
    for (uint64_t sid = (uint64_t)blockIdx.x * 256 + threadIdx.x; sid < n; sid += (uint64_t)gridDim.x * 256) {
        const uint64_t row = dev_of[sid];
        uint64_t r = sid * 0x9E3779B97F4A7C15ull + 1ull;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = regs[row * 4 + k];
            r = hash_mix64(r ^ (((uint64_t)v.y << 32) | v.x));
            r = hash_mix64(r ^ (((uint64_t)v.w << 32) | v.z));
        }
        hr += r;
        const uint64_t a = (uint64_t)__double_as_longlong(ksum[row]), b = (uint64_t)__double_as_longlong(kerr[row]);
        hk += hash_mix64(hash_mix64((sid + 0x632BE59BD9B4E019ull) ^ a) ^ b);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        hr += __shfl_down(hr, off);
        hk += __shfl_down(hk, off);
    }
    const unsigned long long v[4] = {hr, hk, 0, 0};
    block_add_counters(out, v, 0x3u);
}

// scatter/gather between device order and ascending-NodeID order (debug exports)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4 *regs, const uint32_t *dev_of, uint64_t n,
                                                          uint4 *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t sid = t >> 2;
    if (sid >= n) return;
    out[t] = regs[(uint64_t)dev_of[sid] * 4 + (t & 3)];
}

} // namespace hbk
