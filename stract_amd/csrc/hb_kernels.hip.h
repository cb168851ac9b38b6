// hb_kernels.hip.h - gfx950 device code of the HyperBall pass.
//
// What is computed (reference file:line):
//   * counter merge  new[v] = max(old[v], max_{u->v} old[u]), per-register u8 max
//       update_all_counters  crates/core/src/webgraph/centrality/harmonic.rs:116-157
//       HyperLogLog::merge   crates/core/src/hyperloglog.rs:4531-4535
//   * changed detection "any from > to" (harmonic.rs:137-141) == new[v] != old[v]
//   * cardinality estimate HyperLogLog::size (hyperloglog.rs:4484-4516) incl. the
//     6-nearest-neighbour bias lookup (:4407-4470) and linear counting (:4472-4476)
//   * per-node harmonic increment  update_centralities (harmonic.rs:159-176) with
//     KahanSum += (kahan_sum.rs:47-54)
//
// Mapping to the machine: one QUAD (4 lanes) owns one row; lane q of the quad holds
// registers [16q, 16q+16) of the 64-byte counter as a uint4, so a counter gather is one
// 64-byte contiguous segment per quad and one global_load_dwordx4 per lane; a wave64
// covers 16 rows, a 256-thread block 64.  Byte-wise max is done with v_pk_max_u16 on the
// even/odd bytes.  Source indices of a row are loaded 4 at a time (one per lane) and
// broadcast inside the quad with DPP quad_perm, so control flow stays quad-uniform.
// Integer/bitwise work only: no MFMA.  Compiled with -ffp-contract=off: the f64 estimator
// and the Kahan update must round exactly like the reference's scalar Rust.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hb_regs.hip.h"

namespace hbk {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kTableLen = 159;

// Streams that are read or written exactly once per pass (row pointers, source indices, per-row Kahan / size
// words, the freshly written counters): with HB_STREAM_NT they bypass-hint the caches (nontemporal), leaving the
// L2 to the gathered counters.  Experiment switch, see profiles/r02*_stream_nt*.
#ifndef HB_STREAM_NT
#define HB_STREAM_NT 0
#endif
template <class T>
__device__ __forceinline__ T ld_stream(const T *p)
{
#if HB_STREAM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T>
__device__ __forceinline__ void st_stream(T *p, const T &v)
{
    *p = v;
}
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 &v)
{
#if HB_STREAM_NT
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (u32x4 *)p);
#else
    *p = v;
#endif
}
// Per-pass counters are striped: kStripes copies of 4 words, a block adds to stripe blockIdx % kStripes
// (one same-address atomic stream sustains only ~90 updates/us; 8192 waves finishing together made a
// 0.1 ms tail).  The host sums the stripes.
constexpr int kStripes = 64;
constexpr int kCounterWords = 4 * kStripes;

// block-level sum of up to 4 per-wave values (lane 0 of each wave holds its wave's total), then one
// atomic per block and word into the block's stripe
__device__ __forceinline__ void block_add_counters(unsigned long long *counters, const unsigned long long v[4], unsigned mask)
{
    __shared__ unsigned long long s_part[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_part[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4 && ((mask >> threadIdx.x) & 1u)) {
        const int k = threadIdx.x;
        const unsigned long long t = s_part[0][k] + s_part[1][k] + s_part[2][k] + s_part[3][k];
        if (t) atomicAdd(&counters[(blockIdx.x & (kStripes - 1)) * 4 + k], t);
    }
}

struct PassParams {
    const uint64_t *row_ptr;
    const uint32_t *src;
    const uint16_t *src_jp;   // pass 0 only, parallel to src: register index | value << 8 of the source's INITIAL counter
    const uint4 *rd;          // counters of the previous pass ("old"), n_pad rows
    uint4 *wr;                // counters being produced ("new")
    uint4 *part;              // virtual (hub-chunk) rows, indexed by vid - n_pad
    const uint32_t *bits_rd;  // changed bits: real rows = previous pass, virtual rows = this pass
    uint32_t *bits_wr;        // changed bits of this pass for real rows (next frontier)
    uint32_t *kdirty;         // per real row: the last Kahan update moved (sum, err): `+= 0.0` is not yet a no-op
    double *ksum;
    double *kerr;
    uint64_t *size;           // cached size() of rd[row]
    unsigned long long *counters; // kStripes x { [0] changed rows, [1] active edges, [2] rows processed,
                                  // [3] out-degree sum of the changed rows = active edges of the NEXT pass }
    const uint32_t *outdeg;   // per node row: (global) out-degree
    const double *raw;        // HLL64_RAW_ESTIMATE (global copy, staged to LDS)
    const double *bias;       // HLL64_BIAS
    const uint8_t *lc;        // linear-counting table, 65 entries (index = zero registers)
    uint64_t row_lo, row_hi;  // rows of this launch (row_lo multiple of 64)
    // XCD-affine launch (level-1 hub chunks): workgroup b runs on XCD b % 8 (observed dispatch rule; used for
    // speed only) and takes its tiles from group b % 8 = rows [xcd_lo[b % 8], xcd_hi[b % 8])
    int xcd_map;
    uint64_t xcd_lo[8], xcd_hi[8];
    uint64_t n, n_pad;
    uint64_t slice_lo, slice_hi; // rows whose Kahan state this rank owns
    double t_plus_1;          // (t + 1) as f64, harmonic.rs:174
    // edge partition + HB_FLAG_CHANGED_ONLY: the unfused node-row launch records which rows its LOCAL merge changed
    // (lbits); after the union over the ranks only those rows are exchanged, and the epilogue visits only them (ubits)
    uint32_t *lbits;
    const uint32_t *ubits;
};

// ---- HyperLogLog<64>::size(), one quad per counter -------------------------------------
// slice::binary_search_by of Rust >= 1.82 (see oracle/hb_oracle.c)
__device__ __forceinline__ int bias_first_index(const double *raw, double e)
{
    int size = kTableLen, base = 0;
    while (size > 1) {
        int half = size >> 1;
        int mid = base + half;
        if (!(raw[mid] > e)) base = mid;
        size -= half;
    }
    int i = (raw[base] == e) ? base : base + (raw[base] < e ? 1 : 0);
    return i == kTableLen ? kTableLen - 1 : i; // hyperloglog.rs:4413-4416
}

// estimate_bias, hyperloglog.rs:4407-4470 (K = 6 nearest neighbours, mean of their biases)
__device__ __forceinline__ double estimate_bias(const double *raw, const double *bias, double e)
{
    int left = bias_first_index(raw, e);
    int right = (left < kTableLen - 1) ? left + 1 : -1;
    double s = 0.0;
#pragma unroll 1
    for (int k = 0; k < 6; k++) {
        bool take_right;
        if (left >= 0 && right >= 0) {
            double dl = fabs(raw[left] - e), dr = fabs(raw[right] - e);
            take_right = dr < dl;
        } else {
            take_right = left < 0;
        }
        int idx = take_right ? right : left;
        s += bias[idx];
        if (take_right) right = (idx < kTableLen - 1) ? idx + 1 : -1;
        else left = (idx > 0) ? idx - 1 : -1;
    }
    return s / 6.0;
}

__device__ __forceinline__ uint64_t f64_as_usize(double x) // Rust `as usize`
{
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)x;
}

__device__ __forceinline__ double pow2_neg(uint32_t r) // ONE_OVER_POWER_OF_TWO[r], :4043
{
    return __hiloint2double((int)((1023u - r) << 20), 0);
}

// HyperLogLog<64>::size() in two halves, so that the f64 half can run once per ROW instead of once per lane of the
// row's quad (pass_kernel collects the integer halves of four tiles and evaluates 64 distinct rows per wave).
//
// First half, all 4 lanes of the quad call it with their uint4 and all get the same result: sum = sum of 2^-r over the
// 64 registers (the left fold of hyperloglog.rs:4488-4492 is exact and order-independent in f64 when every register
// is <= 47: all partial sums are multiples of 2^-47 below 2^7), zeros = number of zero registers, big = some register
// is > 47 (then the fold must be replayed in register order: hll_fold_quad).
__device__ __forceinline__ void hll_sum_quad(const uint4 &v, double &sum_out, uint32_t &zeros_out, uint32_t &big_out)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    // The 16 terms 2^-r of this lane are added as doubles built from their exponent field (hi word = (1023 - r) << 20):
    // with every register <= 47 all partial sums are multiples of 2^-47 below 2^7, so these additions are exact in any
    // order - the same value as the reference's left fold; four v_add_f64 per word instead of 64-bit integer shifts / adds.
    double acc = 0.0;
    uint32_t zeros = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t r = (w[k] >> (8 * b)) & 0xFFu;
            acc += __hiloint2double((int)((1023u << 20) - (r << 20)), 0); // r <= 255: the exponent field stays positive
        }
        // zero bytes of the word: bit 7 of every byte of z marks a zero byte
        const uint32_t z = ~(((w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[k] | 0x7F7F7F7Fu);
        zeros += __popc(z);
        // largest register of the word, as max over the 16-bit lanes' high bytes and low bytes
        mx = pkmax(mx, pkmax(w[k] & 0x00FF00FFu, (w[k] >> 8) & 0x00FF00FFu));
    }
    uint32_t big = ((mx & 0xFFFFu) > 47u || (mx >> 16) > 47u) ? 1u : 0u;
    // quad reduction (xor 1, xor 2); the f64 sums stay exact for the same reason
    {
        uint32_t lo = (uint32_t)__double2loint(acc), hi = (uint32_t)__double2hiint(acc);
        acc += __hiloint2double((int)quad_perm<0xB1>(hi), (int)quad_perm<0xB1>(lo));
        zeros += quad_perm<0xB1>(zeros);
        big |= quad_perm<0xB1>(big);
        lo = (uint32_t)__double2loint(acc); hi = (uint32_t)__double2hiint(acc);
        acc += __hiloint2double((int)quad_perm<0x4E>(hi), (int)quad_perm<0x4E>(lo));
        zeros += quad_perm<0x4E>(zeros);
        big |= quad_perm<0x4E>(big);
    }
    sum_out = acc; // when big != 0 the value is unused - the fold is replayed in register order
    zeros_out = zeros;
    big_out = big;
}

// The rare case (a register > 47 needs a hash with > 46 leading zeros): the reference's sequential f64 fold over all
// 64 registers in index order; all 4 lanes of the quad call it.
__device__ __forceinline__ double hll_fold_quad(const uint4 &v)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t all[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        all[0 + k] = quad_bcast<0>(w[k]);
        all[4 + k] = quad_bcast<1>(w[k]);
        all[8 + k] = quad_bcast<2>(w[k]);
        all[12 + k] = quad_bcast<3>(w[k]);
    }
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) sum += pow2_neg((all[k] >> (8 * b)) & 0xFFu);
    }
    return sum;
}

// f64 half (hyperloglog.rs:4494-4515) from sum = sum_i 2^-reg[i] and the number of zero registers; any lane, any row.
// raw/bias/lc: tables (LDS or global).
__device__ __forceinline__ uint64_t hll_size_from(double sum, uint32_t zeros, const double *raw, const double *bias, const uint8_t *lc)
{
    const double z = 1.0 / sum;                 // :4494
    const double e = (0.709 * 4096.0) * z;      // :4496  am() * m.powi(2) * z
    double e_star = e;
    if (e <= 320.0) e_star = e - estimate_bias(raw, bias, e); // :4498-4502
    // :4504-4515 : linear counting wins iff v != 0 and 64 ln(64/v) <= 40
    uint32_t l = lc[zeros]; // zeros in 0..64
    if (zeros != 0 && l != 0xFFu) return (uint64_t)l;
    return f64_as_usize(e_star);
}

// All 4 lanes of the quad call this with their uint4; all get the same result.
__device__ __forceinline__ uint64_t hll_size_quad(const uint4 &v, const double *raw, const double *bias,
                                                  const uint8_t *lc)
{
    double sum;
    uint32_t zeros, big;
    hll_sum_quad(v, sum, zeros, big);
    if (big) sum = hll_fold_quad(v); // quad-uniform branch
    return hll_size_from(sum, zeros, raw, bias, lc);
}

// update_centralities for one node (harmonic.rs:159-176) + KahanSum::add_assign.
// Returns whether (sum, err) moved bitwise.  The reference applies this to every node in every
// pass, `+= 0.0` included; a `+= 0.0` that leaves the state bitwise unchanged is a fixed point
// (same inputs next pass), so such a node can be left alone until its counter changes again.
// ("err != 0" is NOT that test: a compensation below half an ulp of sum survives every flush.)
__device__ __forceinline__ bool kahan_update(double &sum, double &err, uint64_t sz_new, uint64_t sz_old,
                                             double t_plus_1)
{
    uint64_t d = (sz_new >= sz_old) ? sz_new - sz_old : 0; // checked_sub().unwrap_or_default()
    double rhs = (double)d / t_plus_1;
    double y = rhs - err;
    double t = sum + y;
    double e = (t - sum) - y;
    const bool moved = (__double_as_longlong(t) != __double_as_longlong(sum)) ||
                       (__double_as_longlong(e) != __double_as_longlong(err));
    err = e;
    sum = t;
    return moved;
}

// ---- the dense pass kernel ---------------------------------------------------------------
// Every source of every row is gathered and every row is written (passes in which most sources changed; the bitmap
// passes - gather only the sources whose changed bit is set - are frontier_kernel below, the data-driven tail the
// sweep kernels).
// REAL      rows are nodes (self = rd[row], output = wr[row]); else virtual hub-chunk rows
//           (output = part[row - n_pad]: the maximum over all current sources dominates the stored partial)
// FUSED     REAL only: estimator + Kahan in the same kernel (single GPU)
// STATS     count active edges / processed rows
// INIT      pass 0, dense only: every real source's counter is still HyperLogLog::default() + add(id) - ONE register
//           set (harmonic.rs:60-62) - so instead of gathering 64 bytes at random the row streams 2 bytes per edge
//           (src_jp, written once at load time) and rebuilds the block in registers; virtual sources are gathered
//           as always.  Same maxima, same bits.
// EPI4      dense fused node rows only: the estimator's f64 half and the Kahan update are deferred until four tiles
//           (64 rows) are merged and then run ONCE PER ROW, lane (g, q) taking row g of the q-th pending tile, instead
//           of four times redundantly per quad; same arithmetic per row, same bits.
template <bool REAL, bool FUSED, bool STATS, int UNROLL, bool INIT = false, bool EPI4 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((REAL && UNROLL >= 4) ? 3 : 4))) void pass_kernel(const PassParams p)
{
    __shared__ double s_raw[FUSED ? kTableLen : 1];
    __shared__ double s_bias[FUSED ? kTableLen : 1];
    __shared__ uint8_t s_lc[68];
    if (FUSED) {
        for (int i = threadIdx.x; i < kTableLen; i += 256) {
            s_raw[i] = p.raw[i];
            s_bias[i] = p.bias[i];
        }
        if (threadIdx.x < 65) s_lc[threadIdx.x] = p.lc[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 2, q = lane & 3;
    const int qshift = lane & ~3;
    // tile -> workgroup map: plain grid stride, or (hub chunks) per-XCD queues
    uint64_t row_lo = p.row_lo, row_hi = p.row_hi, tile0 = blockIdx.x, tstride = gridDim.x;
    if (!REAL && p.xcd_map) {
        const int x = blockIdx.x & 7;
        row_lo = p.xcd_lo[x];
        row_hi = p.xcd_hi[x];
        tile0 = blockIdx.x >> 3;
        tstride = gridDim.x >> 3; // the grid is a multiple of 8
    }
    const uint64_t ntiles = (row_hi - row_lo + 63) >> 6;
    unsigned long long cnt_changed = 0, cnt_active = 0, cnt_rows = 0, cnt_out = 0;

    // Software pipeline over the tiles of this workgroup: the row pointers / out-degree / own counter of the
    // NEXT tile are requested before the gathers of the current one, and the Kahan/size words of the current
    // rows are requested at the top of the iteration - the per-tile chain of dependent memory round trips
    // (row_ptr -> index -> gather -> self -> size/ksum/kerr) shrinks to (index -> gather); node rows have ~5
    // sources each, so that chain, not bandwidth, bounded the node-row kernel.
    constexpr bool kDenseReal = REAL;
    constexpr bool kEpi4 = EPI4 && kDenseReal && FUSED;
    // pass 0 (INIT): per wave 16 scratch counters of 64 x u32 (one per row of the tile), register r of row g at word
    // (r + 4 g) & 63 of the row - the rotation spreads the lanes' 16-byte read-backs over all LDS banks
    __shared__ uint4 s_init4[INIT ? 4 * 16 * 16 : 1];
    uint32_t *init_row = (uint32_t *)s_init4 + (INIT ? (wave * 16 + g) * 64 : 0);
    bool init_used = false;
    if (INIT) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_init4[(wave * 16 + g) * 16 + ((4 * q + k + g) & 15)] = make_uint4(0, 0, 0, 0);
    }
    // deferred epilogue (kEpi4): per wave the first row / Kahan-dirty word of the pending tiles, per lane ITS pending row
    __shared__ uint64_t s_prow16[kEpi4 ? 4 : 1][4];
    __shared__ uint32_t s_pkd16[kEpi4 ? 4 : 1][4];
    uint64_t p_row = 0, p_szfull = 0, p_sz = 0;
    double p_sum = 0.0, p_ks = 0.0, p_ke = 0.0;
    uint32_t p_zeros = 0, p_flags = 0; // 1 = row exists, 2 = changed, 4 = Kahan-dirty, 8 = a register > 47 (p_szfull holds size())
    int npend = 0;                     // pending tiles of this wave, 0..3
    auto flush_pending = [&]() {
        bool err_nz = false;
        if (q < npend && (p_flags & 1u) && (p_flags & 6u)) {
            const uint64_t sz_old = p_sz;
            uint64_t sz_new = sz_old;
            if (p_flags & 2u) sz_new = (p_flags & 8u) ? p_szfull : hll_size_from(p_sum, p_zeros, s_raw, s_bias, s_lc);
            double ks = p_ks, ke = p_ke;
            err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1); // still moving -> visit again
            if (err_nz) {
                p.ksum[p_row] = ks;
                p.kerr[p_row] = ke;
            }
            if (p_flags & 2u) p.size[p_row] = sz_new;
        }
        const uint64_t bal = __ballot(err_nz); // bit 4g + k = row g of pending tile k
        if (lane == 0) {
            for (int k = 0; k < npend; k++) {
                const uint32_t nk16 = pack16(bal & (0x1111111111111111ull << k));
                const uint64_t r16 = s_prow16[kEpi4 ? wave : 0][k];
                const uint32_t kd16 = s_pkd16[kEpi4 ? wave : 0][k];
                if (nk16 | kd16) ((uint16_t *)p.kdirty)[r16 >> 4] = (uint16_t)nk16;
            }
        }
        npend = 0;
    };
    uint64_t nbeg = 0, nend = 0;
    uint32_t nod = 0;
    uint4 nself = make_uint4(0, 0, 0, 0);
    {
        const uint64_t r0 = row_lo + (tile0 << 6) + ((uint64_t)wave << 4) + (uint64_t)g;
        if (tile0 < ntiles && r0 < row_hi) {
            nbeg = ld_stream(&p.row_ptr[r0]);
            nend = ld_stream(&p.row_ptr[r0 + 1]);
            if (REAL && FUSED) nod = ld_stream(&p.outdeg[r0]);
            if (kDenseReal) nself = p.rd[r0 * 4 + q];
        }
    }
    for (uint64_t tile = tile0; tile < ntiles; tile += tstride) {
        const uint64_t row16 = row_lo + (tile << 6) + ((uint64_t)wave << 4); // first row of this wave
        const uint64_t row = row16 + (uint64_t)g;
        const bool valid = row < row_hi;
        const uint64_t beg = nbeg, end = nend;
        const uint32_t od = nod; // out-degree of the row's node, used in the epilogue
        uint4 selfv = nself;
        {   // requests for the next tile of this workgroup
            const uint64_t nrow = row + (tstride << 6);
            nbeg = nend = 0;
            nod = 0;
            nself = make_uint4(0, 0, 0, 0);
            if (tile + tstride < ntiles && nrow < row_hi) {
                nbeg = ld_stream(&p.row_ptr[nrow]);
                nend = ld_stream(&p.row_ptr[nrow + 1]);
                if (REAL && FUSED) nod = ld_stream(&p.outdeg[nrow]);
                if (kDenseReal) nself = p.rd[nrow * 4 + q];
            }
        }
        // dense fused node rows: 4 of 5 rows change, so the estimator/Kahan words are requested now, unconditionally
        uint64_t pre_sz = 0;
        double pre_ks = 0.0, pre_ke = 0.0;
        if (kDenseReal && FUSED && valid) {
            if (kEpi4) {
                if (q == npend) { // this lane owns the row's deferred epilogue
                    pre_sz = ld_stream(&p.size[row]);
                    pre_ks = ld_stream(&p.ksum[row]);
                    pre_ke = ld_stream(&p.kerr[row]);
                }
            } else {
                pre_sz = ld_stream(&p.size[row]);
                if (q == 0) {
                    pre_ks = ld_stream(&p.ksum[row]);
                    pre_ke = ld_stream(&p.kerr[row]);
                }
            }
        }
        Acc acc;
        acc_zero(acc);
        // node rows: self (prefetched) is always merged.  Hub chunks: the maximum over ALL current sources already
        // dominates the stored partial (counters only grow), so the partial is overwritten without being read, and no
        // changed bit is kept (nobody tests it in a dense pass).
        if (kDenseReal && valid) acc_merge(acc, selfv);
        if (beg < end) {
            // all sources of one row are of one kind: real nodes (read rd) or virtual rows (read part)
            const uint32_t first = p.src[beg];
            const uint4 *base = (first >= p.n_pad) ? (const uint4 *)(p.part - p.n_pad * 4) : p.rd;
            const bool real_src = first < p.n_pad;
            for (uint64_t e = beg; e < end; e += 4 * UNROLL) {
                uint32_t idx[UNROLL];
                if (INIT && real_src) {
                    // pass 0: the sources' single registers come with the edge list.  Each lane max-accumulates ITS sources
                    // into the row's 64 x u32 scratch counter in LDS (one ds_max_u32 per source) - nothing is broadcast
                    // to the other lanes of the quad and no lane tests registers that are not its own
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        const uint64_t ee = e + 4 * u + q;
                        idx[u] = (ee < end) ? (uint32_t)ld_stream(&p.src_jp[ee]) : 0u; // value 0 = nothing to merge
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        const uint32_t v = idx[u];
                        if (v >> 8) __hip_atomic_fetch_max(&init_row[((v & 63u) + 4u * (uint32_t)g) & 63u], v >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    init_used = true;
                    continue;
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    uint64_t ee = e + 4 * u + q;
                    idx[u] = (ee < end) ? ld_stream(&p.src[ee]) : kNone;
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    if (STATS && real_src) cnt_active += (idx[u] != kNone);
                }
                {
                    // branch-free: out-of-row slots re-read the row's first source
                    uint4 r[UNROLL][4];
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        uint32_t s0 = quad_bcast<0>(idx[u]), s1 = quad_bcast<1>(idx[u]);
                        uint32_t s2 = quad_bcast<2>(idx[u]), s3 = quad_bcast<3>(idx[u]);
                        s0 = (s0 != kNone) ? s0 : first;
                        s1 = (s1 != kNone) ? s1 : first;
                        s2 = (s2 != kNone) ? s2 : first;
                        s3 = (s3 != kNone) ? s3 : first;
                        r[u][0] = base[(uint64_t)s0 * 4 + q];
                        r[u][1] = base[(uint64_t)s1 * 4 + q];
                        r[u][2] = base[(uint64_t)s2 * 4 + q];
                        r[u][3] = base[(uint64_t)s3 * 4 + q];
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
                        for (int j = 0; j < 4; j++) acc_merge(acc, r[u][j]);
                    }
                }
            }
        }
        if (INIT) {
            // the rows' scratch counters -> register blocks (lane q: registers 16 q .. 16 q + 15), scratch cleared for the next tile
            if (__ballot(init_used)) { // wave-uniform: some row of this tile streamed real sources
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t wv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint4 *cell = &s_init4[(wave * 16 + g) * 16 + ((4 * q + k + g) & 15)]; // registers 16 q + 4 k .. + 3
                    const uint4 c = *cell;
                    *cell = make_uint4(0, 0, 0, 0);
                    wv[k] = c.x | (c.y << 8) | (c.z << 16) | (c.w << 24);
                }
                acc_merge(acc, make_uint4(wv[0], wv[1], wv[2], wv[3]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                init_used = false;
            }
        }
        // ---- row epilogue (quad-uniform decisions come from ballots)
        const uint32_t prev16 = REAL ? (uint32_t)((const uint16_t *)p.bits_rd)[row16 >> 4] : 0u;
        const uint32_t kd16 = (REAL && FUSED) ? (uint32_t)((const uint16_t *)p.kdirty)[row16 >> 4] : 0u;
        const bool self_prev = (prev16 >> g) & 1u;
        const bool kd = (kd16 >> g) & 1u;
        const bool need = valid;
        (void)self_prev;
        const uint4 accv = acc_value(acc);
        const bool lane_diff = need && (!REAL || u4_ne(accv, selfv));
        const uint64_t bal = __ballot(lane_diff);
        const bool changed = ((bal >> qshift) & 0xFull) != 0;
        if (REAL) {
            // lazy double buffer: wr[row] already holds the right value unless the row changed
            // in this or in the previous pass
            if (need) st_stream(&p.wr[row * 4 + q], accv);
        } else {
            if (changed) st_stream(&p.part[(row - p.n_pad) * 4 + q], accv);
        }
        const uint32_t ch16 = pack16(bal);
        if (FUSED) { // changed bits of the node rows -> next frontier (nobody tests a virtual row's bit in a dense pass)
            if (lane == 0 && row16 < row_hi) ((uint16_t *)p.bits_wr)[row16 >> 4] = (uint16_t)ch16;
        }
        if (REAL && !FUSED && p.lbits && lane == 0 && row16 < row_hi) ((uint16_t *)p.lbits)[row16 >> 4] = (uint16_t)ch16;
        if (REAL && FUSED) {
            cnt_changed += __popc(ch16);
            if (changed && q == 0) cnt_out += od;
        }
        if (kEpi4) {
            double rsum;
            uint32_t zeros, big;
            hll_sum_quad(accv, rsum, zeros, big);
            uint64_t szfull = 0;
            if (big) szfull = hll_size_from(hll_fold_quad(accv), zeros, s_raw, s_bias, s_lc); // quad-uniform branch (rare)
            if (q == npend) {
                p_row = row;
                p_sum = rsum;
                p_zeros = zeros;
                p_szfull = szfull;
                p_sz = pre_sz;
                p_ks = pre_ks;
                p_ke = pre_ke;
                p_flags = (need ? 1u : 0u) | (changed ? 2u : 0u) | (kd ? 4u : 0u) | (big ? 8u : 0u);
            }
            if (lane == 0) {
                s_prow16[kEpi4 ? wave : 0][npend] = row16;
                s_pkd16[kEpi4 ? wave : 0][npend] = row16 < row_hi ? kd16 : 0u;
            }
            if (++npend == 4) flush_pending();
        }
        if (REAL && FUSED && !kEpi4) {
            bool err_nz = false;
            if (need && (changed || kd)) {
                const uint64_t sz_old = kDenseReal ? pre_sz : p.size[row];
                const uint64_t sz_new = changed ? hll_size_quad(accv, s_raw, s_bias, s_lc) : sz_old;
                if (q == 0) {
                    double ks = kDenseReal ? pre_ks : p.ksum[row], ke = kDenseReal ? pre_ke : p.kerr[row];
                    err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1); // still moving -> visit again
                    if (err_nz) {
                        p.ksum[row] = ks;
                        p.kerr[row] = ke;
                    }
                    if (changed) p.size[row] = sz_new;
                }
            }
            const uint32_t nk16 = pack16(__ballot(err_nz));
            if (lane == 0 && row16 < row_hi && (nk16 | kd16)) ((uint16_t *)p.kdirty)[row16 >> 4] = (uint16_t)nk16;
        }
    }
    if (kEpi4 && npend) flush_pending();
    // ---- totals: wave -> block -> one atomic per word into the block's counter stripe
    if (REAL || STATS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cnt_out += __shfl_down(cnt_out, off);
            if (STATS) cnt_active += __shfl_down(cnt_active, off);
        }
        // cnt_changed is identical in all lanes of the wave (derived from a ballot)
        const unsigned long long v[4] = {cnt_changed, cnt_active, cnt_rows, cnt_out};
        block_add_counters(p.counters, v, (REAL ? 0x9u : 0u) | (STATS ? 0x2u : 0u));
    }
}

// ---- the bitmap (frontier) pass, restructured ------------------------------------------------------------------------
// A source is gathered only if its changed bit is set (results-inert); rows nothing happened to are
// left alone (lazy double buffer).  Built for what bounds that pass: with few active
// sources it is a chain of DEPENDENT round trips per row - index -> changed-bit word -> counter gather, repeated for every
// 16 sources, then the row's own counter - at a handful of waves per SIMD, not bytes.  Here a quad takes ALL (<= 64)
// indices of its row in one go (16 per lane), then all their bit words, then issues only the gathers that are needed,
// together with the row's own counter: three round trips per row instead of up to fourteen.  Gather slots in which no
// quad of the wave has an active source are skipped altogether (wave-uniform test on a ballot).
// W = index slots per lane and batch: 16 (64 sources per quad: hub chunks) or 4 (16 sources: node rows have ~5)
// Two filters in front of the bitmap test were built and measured SLOWER (DESIGN.md "tried and rejected", round 3): an
// LDS-staged coarse summary of the bitmap, and a "hot prefix" shortcut (sources below the first changed segment decided by
// <= 8 register compares): the bit tests they save are L1/L2 hits that overlap with the rest of the row.
template <bool REAL, bool FUSED, bool STATS, int W>
__global__ __launch_bounds__(256) void frontier_kernel(const PassParams p)
{
    __shared__ double s_raw[FUSED ? kTableLen : 1];
    __shared__ double s_bias[FUSED ? kTableLen : 1];
    __shared__ uint8_t s_lc[68];
    if (FUSED) {
        for (int i = threadIdx.x; i < kTableLen; i += 256) {
            s_raw[i] = p.raw[i];
            s_bias[i] = p.bias[i];
        }
        if (threadIdx.x < 65) s_lc[threadIdx.x] = p.lc[threadIdx.x];
    }
    if (FUSED) __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 2, q = lane & 3;
    const int qshift = lane & ~3;
    uint64_t row_lo = p.row_lo, row_hi = p.row_hi, tile0 = blockIdx.x, tstride = gridDim.x;
    if (!REAL && p.xcd_map) {
        const int x = blockIdx.x & 7;
        row_lo = p.xcd_lo[x];
        row_hi = p.xcd_hi[x];
        tile0 = blockIdx.x >> 3;
        tstride = gridDim.x >> 3; // the grid is a multiple of 8
    }
    const uint64_t ntiles = (row_hi - row_lo + 63) >> 6;
    unsigned long long cnt_changed = 0, cnt_active = 0, cnt_rows = 0, cnt_out = 0;
    // row pointers (and the changed / Kahan-dirty words) of the NEXT tile are requested one iteration ahead
    uint64_t nbeg = 0, nend = 0;
    uint32_t nprev16 = 0, nkd16 = 0;
    auto request = [&](uint64_t tile) {
        nbeg = nend = 0;
        nprev16 = nkd16 = 0;
        const uint64_t r16 = row_lo + (tile << 6) + ((uint64_t)wave << 4), r = r16 + (uint64_t)g;
        if (tile < ntiles && r < row_hi) {
            nbeg = ld_stream(&p.row_ptr[r]);
            nend = ld_stream(&p.row_ptr[r + 1]);
        }
        if (REAL && tile < ntiles && r16 < row_hi) {
            nprev16 = (uint32_t)((const uint16_t *)p.bits_rd)[r16 >> 4];
            if (FUSED) nkd16 = (uint32_t)((const uint16_t *)p.kdirty)[r16 >> 4];
        }
    };
    request(tile0);
    for (uint64_t tile = tile0; tile < ntiles; tile += tstride) {
        const uint64_t row16 = row_lo + (tile << 6) + ((uint64_t)wave << 4); // first row of this wave
        const uint64_t row = row16 + (uint64_t)g;
        const bool valid = row < row_hi;
        const uint64_t beg = nbeg, end = nend;
        const uint32_t prev16 = nprev16, kd16 = nkd16;
        request(tile + tstride);
        const bool self_prev = (prev16 >> g) & 1u;
        const bool kd = (kd16 >> g) & 1u;
        const uint4 *selfp = REAL ? (p.rd + row * 4 + q) : (p.part + (row - p.n_pad) * 4 + q);
        // node rows that must be written anyway (changed last pass / Kahan still moving): their own counter is requested now
        uint4 selfv = make_uint4(0, 0, 0, 0);
        const bool early_self = REAL && valid && (self_prev || kd);
        if (early_self) selfv = *selfp;
        Acc acc;
        acc_zero(acc);
        bool lane_act = false;
        for (uint64_t e0 = beg; e0 < end; e0 += 4 * W) { // hub chunks: one iteration unless hb_options.chunk > 64
            // ---- round trip 1: all indices of the batch, W per lane (slot j of lane q = source e0 + 4 j + q)
            const uint64_t span = end - e0;
            uint32_t idx[W];
            uint64_t bal4[W / 4];
#pragma unroll
            for (int b = 0; b < W / 4; b++) bal4[b] = __ballot(span > (uint64_t)(16 * b));
#pragma unroll
            for (int b = 0; b < W / 4; b++) {
                if (bal4[b]) { // wave-uniform: some row of the wave reaches this quarter
#pragma unroll
                    for (int j = 4 * b; j < 4 * b + 4; j++) {
                        const uint64_t ee = e0 + 4 * j + q;
                        idx[j] = (ee < end) ? ld_stream(&p.src[ee]) : kNone;
                    }
                } else {
#pragma unroll
                    for (int j = 4 * b; j < 4 * b + 4; j++) idx[j] = kNone;
                }
            }
            // all sources of one row are of one kind: real nodes (read rd) or virtual rows (read part)
            const uint32_t first = quad_bcast<0>(idx[0]);
            const bool real_src = first < p.n_pad;
            const uint4 *base = real_src ? p.rd : (const uint4 *)(p.part - p.n_pad * 4);
            // ---- round trip 2: the changed bits of all of them
            uint32_t wb[W];
#pragma unroll
            for (int b = 0; b < W / 4; b++) {
                if (bal4[b]) {
#pragma unroll
                    for (int j = 4 * b; j < 4 * b + 4; j++) wb[j] = (idx[j] != kNone) ? p.bits_rd[idx[j] >> 5] : 0u;
                } else {
#pragma unroll
                    for (int j = 4 * b; j < 4 * b + 4; j++) wb[j] = 0u;
                }
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                if (!((wb[j] >> (idx[j] & 31u)) & 1u)) idx[j] = kNone;
                lane_act |= (idx[j] != kNone);
                if (STATS && real_src) cnt_active += (idx[j] != kNone);
            }
            // a hub chunk reads its stored partial only if something reaches it: known now, requested with the gathers
            if (!REAL && e0 == beg) {
                const bool t0 = ((__ballot(lane_act) >> qshift) & 0xFull) != 0;
                if (valid && t0) selfv = *selfp;
            }
            // ---- round trip 3: the gathers that are needed, two slots (8 sources per quad) at a time
#pragma unroll
            for (int j = 0; j < W; j += 2) {
                if (!__ballot((idx[j] != kNone) | (idx[j + 1] != kNone))) continue; // no quad of the wave has work in these slots
                uint4 r[2][4];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t s0 = quad_bcast<0>(idx[j + u]), s1 = quad_bcast<1>(idx[j + u]);
                    const uint32_t s2 = quad_bcast<2>(idx[j + u]), s3 = quad_bcast<3>(idx[j + u]);
                    r[u][0] = r[u][1] = r[u][2] = r[u][3] = make_uint4(0, 0, 0, 0); // max with 0 = identity
                    if (s0 != kNone) r[u][0] = base[(uint64_t)s0 * 4 + q];
                    if (s1 != kNone) r[u][1] = base[(uint64_t)s1 * 4 + q];
                    if (s2 != kNone) r[u][2] = base[(uint64_t)s2 * 4 + q];
                    if (s3 != kNone) r[u][3] = base[(uint64_t)s3 * 4 + q];
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) acc_merge(acc, r[u][k]);
                }
            }
        }
        // ---- row epilogue: exactly pass_kernel's frontier epilogue
        const bool touched = ((__ballot(lane_act) >> qshift) & 0xFull) != 0;
        if (REAL) cnt_rows += (valid && touched && q == 0); // V_t (hb_pass_stats.touched)
        const bool need = valid && (touched || (REAL && (self_prev || kd)));
        if (need) {
            if (REAL && !early_self) selfv = *selfp; // a node row reached by a changed source only
            acc_merge(acc, selfv);
        }
        const uint4 accv = acc_value(acc);
        const bool lane_diff = need && u4_ne(accv, selfv);
        const uint64_t bal = __ballot(lane_diff);
        const bool changed = ((bal >> qshift) & 0xFull) != 0;
        if (REAL) {
            // lazy double buffer: wr[row] already holds the right value unless the row changed in this or in the previous pass
            if (need && (changed || self_prev)) st_stream(&p.wr[row * 4 + q], accv);
        } else {
            if (changed) st_stream(&p.part[(row - p.n_pad) * 4 + q], accv);
        }
        const uint32_t ch16 = pack16(bal);
        if (FUSED || !REAL) {
            // changed bits: real rows -> next frontier; virtual rows -> this pass' bits
            uint16_t *dst = REAL ? (uint16_t *)p.bits_wr : (uint16_t *)p.bits_rd;
            if (lane == 0 && row16 < row_hi) dst[row16 >> 4] = (uint16_t)ch16;
        }
        if (REAL && !FUSED && p.lbits && lane == 0 && row16 < row_hi) ((uint16_t *)p.lbits)[row16 >> 4] = (uint16_t)ch16;
        if (REAL && FUSED) {
            cnt_changed += __popc(ch16);
            if (changed && q == 0) cnt_out += p.outdeg[row];
            bool err_nz = false;
            if (need && (changed || kd)) {
                const uint64_t sz_old = p.size[row];
                const uint64_t sz_new = changed ? hll_size_quad(accv, s_raw, s_bias, s_lc) : sz_old;
                if (q == 0) {
                    double ks = p.ksum[row], ke = p.kerr[row];
                    err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1); // still moving -> visit again
                    if (err_nz) {
                        p.ksum[row] = ks;
                        p.kerr[row] = ke;
                    }
                    if (changed) p.size[row] = sz_new;
                }
            }
            const uint32_t nk16 = pack16(__ballot(err_nz));
            if (lane == 0 && row16 < row_hi && (nk16 | kd16)) ((uint16_t *)p.kdirty)[row16 >> 4] = (uint16_t)nk16;
        }
    }
    // ---- totals: wave -> block -> one atomic per word into the block's counter stripe
    if (REAL || STATS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cnt_out += __shfl_down(cnt_out, off);
            if (STATS) cnt_active += __shfl_down(cnt_active, off);
            if (REAL) cnt_rows += __shfl_down(cnt_rows, off);
        }
        // cnt_changed is identical in all lanes of the wave (derived from a ballot)
        const unsigned long long v[4] = {cnt_changed, cnt_active, cnt_rows, cnt_out};
        block_add_counters(p.counters, v, (REAL ? 0xDu : 0u) | (STATS ? 0x2u : 0u));
    }
}

// ---- sweep mode (data-driven passes: the mid-tail and the convergence tail) -------------------
// When a minority of the nodes changed in the previous pass, reading and bit-testing every index
// (bitmap frontier pass) costs far more than the work.  The reference switches to
// update_changed_counters (harmonic.rs:75-114: only the out-edges of the exactly-tracked changed set)
// in the same situation.  Here: the transposed work-row graph (out_ptr/out_rows: for every node or
// virtual row, the work rows that read it) turns the changed set into a TOUCH bitmap over the work rows
// (one bit per row; a few MB, cache resident, so the atomic ORs are cheap - 64-bit per-row masks and
// per-level worklists were both measured slower, profiles/r02a_*); every level is then one ordered sweep
// over its slice of that bitmap: a wave takes 64 words (2048 rows), clears them, compacts the set bits into
// a row list in LDS and runs the listed rows, one quad each, with exactly the frontier-mode row semantics
// of pass_kernel - registers / Kahan state / changed bits are bit-identical.  A virtual row that changed
// touches its parent, so changes climb the chunk trees inside the pass.  Rows are visited in ascending
// order (their state arrays are read almost sequentially, unlike worklists filled in arrival order) and the
// bitmap is left all-zero for the next pass.
constexpr uint64_t kHeavyReaders = 4096; // a seed with more readers than this is expanded grid-wide

struct SweepParams {
    PassParams p;
    const uint64_t *out_ptr;   // rows_total + 1
    const uint32_t *out_rows;  // work rows reading each source
    uint32_t *touch;           // 1 bit per work row: has an active source / must be revisited
    uint32_t *seeds;           // nodes changed in the previous pass (capacity n_pad)
    uint32_t *heavy;           // seeds with very long reader lists (expanded by the whole grid)
    unsigned int *counts;      // this pass' slot: [0] seeds, [1] heavy seeds
    unsigned int *counts_next; // the other slot (zeroed by this pass' first kernel for the next sweep pass)
};

__device__ __forceinline__ void touch_set(uint32_t *touch, uint32_t r)
{
    const uint32_t bit = 1u << (r & 31u);
    // pre-test at the L2 (device-coherent load: a row usually has several changed sources, only the first
    // needs the atomic; a stale 0 would only cost a redundant one - bits are never cleared while being set)
    if (!(__hip_atomic_load(&touch[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&touch[r >> 5], bit);
}

// one thread per 32 node rows: nodes that changed in the previous pass become seeds (their readers are
// touched by sweep_expand_kernel).  They and the Kahan-dirty nodes are also revisited themselves, on the
// cheap path of sweep_rows_kernel<true>, which reads those two bitmaps next to the touch bitmap.
__global__ __launch_bounds__(256) void sweep_collect_kernel(const SweepParams sp)
{
    if (blockIdx.x == 0 && threadIdx.x < 2) sp.counts_next[threadIdx.x] = 0; // last used two passes ago
    const uint64_t words = sp.p.n_pad >> 5;
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w0 = (uint64_t)blockIdx.x * 256; w0 < words; w0 += stride) { // wave-uniform trip count
        const uint64_t w = w0 + threadIdx.x;
        const uint32_t ch_in = (w < words) ? sp.p.bits_rd[w] : 0u;
        uint32_t ch = ch_in;
        // wave-aggregated reservation in the seed list
        const uint32_t nch = __popc(ch);
        uint32_t pch = nch; // inclusive prefix sum over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a = __shfl_up(pch, off);
            if (lane >= off) pch += a;
        }
        const uint32_t tot = __shfl(pch, 63);
        uint32_t base = 0;
        if (lane == 0 && tot) base = atomicAdd(&sp.counts[0], tot);
        base = __shfl(base, 0) + pch - nch;
        while (ch) {
            const int b = __ffs((int)ch) - 1;
            ch &= ch - 1;
            sp.seeds[base++] = (uint32_t)(w << 5) + (uint32_t)b;
        }
    }
}

// Seeds -> touch bits.  A wave takes 64 seeds and walks the CONCATENATION of their reader lists 64 entries
// at a time (exclusive prefix sums of the list lengths; every lane finds the seed of its entry by a binary
// search over the lanes' offsets with ds_bpermute), so lanes stay busy whatever the out-degrees are.  Seeds
// with more than kHeavyReaders readers (hubs stay in the changed set longest) go to the grid-wide kernel.
__global__ __launch_bounds__(256) void sweep_expand_kernel(const SweepParams sp)
{
    const int lane = threadIdx.x & 63;
    const uint32_t nseeds = sp.counts[0];
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (uint32_t i0 = wave * 64; i0 < nseeds; i0 += nwaves * 64) { // wave-uniform trip count
        const uint32_t i = i0 + lane;
        uint64_t b = 0, e = 0;
        uint32_t u = 0;
        if (i < nseeds) {
            u = sp.seeds[i];
            b = sp.out_ptr[u];
            e = sp.out_ptr[u + 1];
        }
        const bool is_heavy = e - b > kHeavyReaders;
        const uint64_t hm = __ballot(is_heavy);
        if (hm) {
            uint32_t hb = 0;
            const int leader = __ffsll((long long)hm) - 1;
            if (lane == leader) hb = atomicAdd(&sp.counts[1], (unsigned)__popcll(hm));
            hb = __shfl(hb, leader);
            if (is_heavy) {
                sp.heavy[hb + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = u;
                e = b;
            }
        }
        const uint32_t len = (uint32_t)(e - b);
        uint32_t incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a = __shfl_up(incl, off);
            if (lane >= off) incl += a;
        }
        const uint32_t total = __shfl(incl, 63);
        const uint32_t excl = incl - len;
        const uint32_t blo = (uint32_t)b, bhi = (uint32_t)(b >> 32);
        for (uint32_t r = 0; r < total; r += 64) {
            const uint32_t item = r + lane;
            // owner = last lane whose exclusive offset is <= item (lanes with empty lists share offsets with
            // their successor; the LAST such lane is the one that owns the entry)
            int lo = 0, hi = 64;
#pragma unroll
            for (int step = 0; step < 6; step++) {
                const int mid = (lo + hi) >> 1;
                const uint32_t v = __shfl(excl, mid);
                if (v <= item) lo = mid;
                else hi = mid;
            }
            const uint32_t oex = __shfl(excl, lo);
            const uint64_t ob = ((uint64_t)__shfl(bhi, lo) << 32) | __shfl(blo, lo);
            if (item < total) touch_set(sp.touch, sp.out_rows[ob + (item - oex)]);
        }
    }
}

// Convergence tail (a few thousand changed nodes at most): seed collection and expansion in ONE launch - every lane takes a
// word of the changed bitmap and walks the reader lists of its set bits itself; lists longer than 64 entries are walked by
// the whole wave (a hub that still changes this late is rare but must not serialise on one lane).  No seed list, no counts.
__global__ __launch_bounds__(256) void sweep_seed_small_kernel(const SweepParams sp)
{
    if (blockIdx.x == 0 && threadIdx.x < 2) { // unused here: both slots are left clean for whichever pass collects seeds next
        sp.counts[threadIdx.x] = 0;
        sp.counts_next[threadIdx.x] = 0;
    }
    const uint64_t words = sp.p.n_pad >> 5;
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w0 = (uint64_t)blockIdx.x * 256; w0 < words; w0 += stride) { // wave-uniform trip count
        const uint64_t w = w0 + threadIdx.x;
        uint32_t ch = (w < words) ? sp.p.bits_rd[w] : 0u;
        if (!__ballot(ch != 0)) continue;
        uint32_t lng = 0; // this lane's seeds with long reader lists
        while (ch) {
            const int b = __ffs((int)ch) - 1;
            ch &= ch - 1;
            const uint64_t u = (w << 5) + (uint64_t)b;
            const uint64_t kb = sp.out_ptr[u], ke = sp.out_ptr[u + 1];
            if (ke - kb > 64) lng |= 1u << b;
            else
                for (uint64_t k = kb; k < ke; k++) touch_set(sp.touch, sp.out_rows[k]);
        }
        uint64_t owners;
        while ((owners = __ballot(lng != 0)) != 0) {
            const int src = __ffsll((long long)owners) - 1;
            const uint32_t m = __shfl(lng, src);
            const int b = __ffs((int)m) - 1;
            if (lane == src) lng &= lng - 1;
            const uint64_t u = ((w0 + (uint64_t)(threadIdx.x & ~63) + (uint64_t)src) << 5) + (uint64_t)b;
            const uint64_t kb = sp.out_ptr[u], ke = sp.out_ptr[u + 1];
            for (uint64_t k = kb + lane; k < ke; k += 64) touch_set(sp.touch, sp.out_rows[k]);
        }
    }
}

__global__ __launch_bounds__(256) void sweep_expand_heavy_kernel(const SweepParams sp)
{
    const uint32_t nheavy = sp.counts[1];
    const uint64_t wbase = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64, nthreads = (uint64_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = 0; i < nheavy; i++) {
        const uint32_t u = sp.heavy[i];
        const uint64_t b = sp.out_ptr[u], e = sp.out_ptr[u + 1];
        for (uint64_t k0 = b + wbase; k0 < e; k0 += nthreads) { // wave-uniform trip count
            const uint64_t k = k0 + lane;
            if (k < e) touch_set(sp.touch, sp.out_rows[k]);
        }
    }
}

// the touched rows of [row_lo, row_hi) (multiples of 64), one quad each; REAL: node rows (self = rd[row], fused
// estimator + Kahan), else virtual rows (self = part[row - n_pad]; a changed row touches its readers).
// A wave-iteration takes 64 bitmap words as 16 groups of 4 consecutive words (128 rows) that lie nwaves groups
// apart: touched rows cluster (the readers of late changers are cold chunks / low-degree rows, which the device
// order keeps together), and contiguous 2048-row slabs gave a few waves all the work.  The wave OWNS the rows of
// its words for the whole pass, so their changed / Kahan-dirty words are assembled in LDS and stored once - no
// global atomics and no clearing of those bitmaps (every word of the range is rewritten).
template <bool REAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void sweep_rows_kernel(const SweepParams sp)
{
    __shared__ double s_raw[REAL ? kTableLen : 1];
    __shared__ double s_bias[REAL ? kTableLen : 1];
    __shared__ uint8_t s_lc[68];
    __shared__ uint16_t s_list[4][2048]; // per wave: (owner lane << 5 | bit) of the set bits of its 64 words
    __shared__ uint32_t s_word[4][64];   // bitmap word index loaded by each lane
    __shared__ uint32_t s_chw[4][64];    // changed bits of this pass, per owned word
    __shared__ uint32_t s_kdw[4][64];    // Kahan-dirty bits, per owned word (REAL)
    constexpr int kU = 2;                // index quads per gather round
    const PassParams &p = sp.p;
    if (REAL) {
        for (int i = threadIdx.x; i < kTableLen; i += 256) {
            s_raw[i] = p.raw[i];
            s_bias[i] = p.bias[i];
        }
        if (threadIdx.x < 65) s_lc[threadIdx.x] = p.lc[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane >> 2, q = lane & 3, qshift = lane & ~3;
    uint16_t *list = s_list[wv];
    uint32_t *wordof = s_word[wv], *chw = s_chw[wv], *kdw = s_kdw[wv];
    const uint64_t w_lo = p.row_lo >> 5, w_hi = (p.row_hi + 31) >> 5;
    const uint64_t nwaves = (uint64_t)gridDim.x * 4, wid = (uint64_t)blockIdx.x * 4 + wv;
    const uint64_t ngroups = (w_hi - w_lo + 3) >> 2;
    unsigned long long cnt_changed = 0, cnt_out = 0, cnt_rows = 0;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto wave_scan = [&](uint32_t v, uint32_t &total) { // inclusive prefix sum over the wave
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a = __shfl_up(incl, off);
            if (lane >= off) incl += a;
        }
        total = __shfl(incl, 63);
        return incl;
    };
    for (uint64_t g0 = 0; g0 < ngroups; g0 += 16 * nwaves) { // wave-uniform trip count
        const uint64_t gi = g0 + (uint64_t)(lane >> 2) * nwaves + wid;
        const uint64_t w = w_lo + gi * 4 + (uint64_t)(lane & 3);
        const bool in_range = gi < ngroups && w < w_hi;
        uint32_t word = in_range ? sp.touch[w] : 0u;
        if (word) sp.touch[w] = 0; // consumed: the bitmap is all-zero again after the pass
        // node rows that no changed source reaches but that changed in the previous pass (lazy double buffer:
        // their counter must be carried over to the other buffer) or whose Kahan state is still moving (the
        // reference adds +0.0 to every node in every pass): cheap path below, no index or counter gathers
        const uint32_t pw = (REAL && in_range) ? p.bits_rd[w] : 0u;
        const uint32_t kw = (REAL && in_range) ? p.kdirty[w] : 0u;
        uint32_t cheap = (pw | kw) & ~word;
        uint32_t total = 0;
        const uint32_t incl = wave_scan(__popc(word), total);
        const bool any_cheap = REAL && __ballot(cheap != 0) != 0;
        if (total == 0 && !any_cheap) {
            // nothing to run: the owned words of this pass' changed bitmap still have to be (re)written
            if (in_range) {
                if (REAL) p.bits_wr[w] = 0;
                else ((uint32_t *)p.bits_rd)[w] = 0;
            }
            continue;
        }
        wordof[lane] = (uint32_t)w;
        chw[lane] = 0;
        kdw[lane] = kw;
        {
            uint32_t pos = incl - __popc(word);
            while (word) {
                const int b = __ffs((int)word) - 1;
                word &= word - 1;
                list[pos++] = (uint16_t)((lane << 5) | b);
            }
        }
        wave_sync();
        // software pipeline over the batches of 16 rows: the row pointers and the own counter of the NEXT batch
        // are requested before the gathers of the current one (the chain list -> row_ptr -> index -> bit ->
        // gather -> state is what bounds this kernel, not bandwidth)
        uint64_t nrow = 0, nbeg = 0, nend = 0;
        uint32_t nent = 0;
        uint4 nself = make_uint4(0, 0, 0, 0);
        bool nvalid = (uint32_t)g < total;
        if (nvalid) {
            nent = list[g];
            nrow = ((uint64_t)wordof[nent >> 5] << 5) + (nent & 31u);
            nbeg = p.row_ptr[nrow];
            nend = p.row_ptr[nrow + 1];
            nself = REAL ? p.rd[nrow * 4 + q] : p.part[(nrow - p.n_pad) * 4 + q];
        }
        for (uint32_t base = 0; base < total; base += 16) {
            const bool valid = nvalid;
            const uint64_t row = nrow, beg = nbeg, end = nend;
            const uint32_t ent = nent;
            const uint4 selfv = nself;
            {
                const uint32_t li = base + 16 + (uint32_t)g;
                nvalid = li < total;
                nrow = nbeg = nend = 0;
                nent = 0;
                nself = make_uint4(0, 0, 0, 0);
                if (nvalid) {
                    nent = list[li];
                    nrow = ((uint64_t)wordof[nent >> 5] << 5) + (nent & 31u);
                    nbeg = p.row_ptr[nrow];
                    nend = p.row_ptr[nrow + 1];
                    nself = REAL ? p.rd[nrow * 4 + q] : p.part[(nrow - p.n_pad) * 4 + q];
                }
            }
            Acc acc;
            acc_zero(acc);
            bool lane_act = false;
            if (beg < end) {
                const uint32_t first = p.src[beg];
                const uint4 *srcbase = (first >= p.n_pad) ? (const uint4 *)(p.part - p.n_pad * 4) : p.rd;
                for (uint64_t e = beg; e < end; e += 4 * kU) { // 4 * kU sources per round: indices, bit tests, gathers
                    uint32_t idx[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        const uint64_t ee = e + 4 * u + q;
                        idx[u] = (ee < end) ? p.src[ee] : kNone;
                    }
                    uint32_t wb[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) wb[u] = (idx[u] != kNone) ? p.bits_rd[idx[u] >> 5] : 0u;
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        if (!((wb[u] >> (idx[u] & 31u)) & 1u)) idx[u] = kNone;
                        lane_act |= (idx[u] != kNone);
                    }
                    uint4 r[kU][4];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        const uint32_t s0 = quad_bcast<0>(idx[u]), s1 = quad_bcast<1>(idx[u]);
                        const uint32_t s2 = quad_bcast<2>(idx[u]), s3 = quad_bcast<3>(idx[u]);
                        r[u][0] = r[u][1] = r[u][2] = r[u][3] = make_uint4(0, 0, 0, 0); // max with 0 = identity
                        if (s0 != kNone) r[u][0] = srcbase[(uint64_t)s0 * 4 + q];
                        if (s1 != kNone) r[u][1] = srcbase[(uint64_t)s1 * 4 + q];
                        if (s2 != kNone) r[u][2] = srcbase[(uint64_t)s2 * 4 + q];
                        if (s3 != kNone) r[u][3] = srcbase[(uint64_t)s3 * 4 + q];
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++) {
#pragma unroll
                        for (int j = 0; j < 4; j++) acc_merge(acc, r[u][j]);
                    }
                }
            }
            acc_merge(acc, selfv);
            const uint4 accv = acc_value(acc);
            const uint64_t bal = __ballot(valid && u4_ne(accv, selfv));
            const bool changed = ((bal >> qshift) & 0xFull) != 0;
            const uint32_t owner = ent >> 5, bit = 1u << (ent & 31u);
            if (changed && q == 0) atomicOr(&chw[owner], bit); // LDS
            if (REAL) {
                const bool touched = ((__ballot(lane_act) >> qshift) & 0xFull) != 0;
                cnt_rows += (valid && touched && q == 0);
                const bool self_prev = valid && ((p.bits_rd[row >> 5] >> (row & 31u)) & 1u);
                const bool kd = valid && ((p.kdirty[row >> 5] >> (row & 31u)) & 1u);
                if (valid && (changed || self_prev)) p.wr[row * 4 + q] = accv; // lazy double buffer
                if (changed && q == 0) cnt_out += p.outdeg[row];
                if (valid && (changed || kd)) {
                    const uint64_t sz_old = p.size[row];
                    const uint64_t sz_new = changed ? hll_size_quad(accv, s_raw, s_bias, s_lc) : sz_old;
                    if (q == 0) {
                        double ks = p.ksum[row], ke = p.kerr[row];
                        const bool err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1);
                        if (err_nz) {
                            p.ksum[row] = ks;
                            p.kerr[row] = ke;
                        }
                        if (changed) p.size[row] = sz_new;
                        if (err_nz && !kd) atomicOr(&kdw[owner], bit);  // LDS
                        if (!err_nz && kd) atomicAnd(&kdw[owner], ~bit); // LDS
                    }
                }
            } else if (changed) {
                p.part[(row - p.n_pad) * 4 + q] = accv;
                if (q == 0) { // the readers (normally exactly one parent) must look at this partial
                    for (uint64_t k = sp.out_ptr[row]; k < sp.out_ptr[row + 1]; k++) touch_set(sp.touch, sp.out_rows[k]);
                }
            }
        }
        if (REAL && any_cheap) {
            wave_sync(); // the list is rewritten
            uint32_t total2 = 0;
            const uint32_t incl2 = wave_scan(__popc(cheap), total2);
            uint32_t pos2 = incl2 - __popc(cheap);
            while (cheap) {
                const int b = __ffs((int)cheap) - 1;
                cheap &= cheap - 1;
                list[pos2++] = (uint16_t)((lane << 5) | b);
            }
            wave_sync();
            for (uint32_t base = 0; base < total2; base += 32) { // two rows per quad and round
                uint64_t row2[2];
                uint32_t ent2[2];
                bool sp2[2], kd2[2];
                uint4 cv[2];
                double ks[2], ke[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint32_t li = base + 16 * j + (uint32_t)g;
                    const bool v = li < total2;
                    ent2[j] = v ? (uint32_t)list[li] : 0u;
                    row2[j] = ((uint64_t)wordof[ent2[j] >> 5] << 5) + (ent2[j] & 31u);
                    sp2[j] = v && ((p.bits_rd[row2[j] >> 5] >> (row2[j] & 31u)) & 1u);
                    kd2[j] = v && ((p.kdirty[row2[j] >> 5] >> (row2[j] & 31u)) & 1u);
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    cv[j] = make_uint4(0, 0, 0, 0);
                    ks[j] = ke[j] = 0.0;
                    if (sp2[j]) cv[j] = p.rd[row2[j] * 4 + q];
                    if (kd2[j] && q == 0) {
                        ks[j] = p.ksum[row2[j]];
                        ke[j] = p.kerr[row2[j]];
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (sp2[j]) p.wr[row2[j] * 4 + q] = cv[j]; // unchanged: carried over to the other buffer
                    if (kd2[j] && q == 0) {
                        // update_centralities with size(new) == size(old): `+= 0.0` (harmonic.rs:159-176)
                        const bool moved = kahan_update(ks[j], ke[j], 0, 0, p.t_plus_1);
                        if (moved) {
                            p.ksum[row2[j]] = ks[j];
                            p.kerr[row2[j]] = ke[j];
                        } else {
                            atomicAnd(&kdw[ent2[j] >> 5], ~(1u << (ent2[j] & 31u))); // LDS
                        }
                    }
                }
            }
        }
        wave_sync();
        // the owner lanes store the words of the bitmaps this wave owns
        if (in_range) {
            const uint32_t cw = chw[lane];
            cnt_changed += __popc(cw);
            if (REAL) {
                p.bits_wr[w] = cw;
                if (kdw[lane] != kw) p.kdirty[w] = kdw[lane];
            } else {
                ((uint32_t *)p.bits_rd)[w] = cw; // this pass' changed bits of the virtual rows
            }
        }
        wave_sync(); // LDS arrays are rewritten in the next iteration
    }
    if (REAL) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cnt_out += __shfl_down(cnt_out, off);
            cnt_rows += __shfl_down(cnt_rows, off);
            cnt_changed += __shfl_down(cnt_changed, off);
        }
        const unsigned long long v[4] = {cnt_changed, 0, cnt_rows, cnt_out};
        block_add_counters(p.counters, v, 0xDu);
    }
}

// out-degree histogram of a source list (load time)
__global__ __launch_bounds__(256) void histogram_kernel(const uint32_t *src, uint64_t m, uint32_t *count)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)gridDim.x * 256) atomicAdd(&count[src[i]], 1u);
}

// ---- transposed work-row graph (built once per load) ------------------------------------------
// count[s] = number of work rows reading s; then (after a host-side exclusive scan) fill.
__global__ __launch_bounds__(256) void transpose_count_kernel(const uint64_t *row_ptr, const uint32_t *src, uint64_t rows,
                                                              uint32_t *count)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nq = (uint64_t)gridDim.x * 64; // quads in the grid
    const int q = threadIdx.x & 3;
    for (uint64_t row = t >> 2; row < rows; row += nq) {
        const uint64_t b = row_ptr[row], e = row_ptr[row + 1];
        for (uint64_t k = b + q; k < e; k += 4) atomicAdd(&count[src[k]], 1u);
    }
}
__global__ __launch_bounds__(256) void transpose_fill_kernel(const uint64_t *row_ptr, const uint32_t *src, uint64_t rows,
                                                             const uint64_t *out_ptr, uint32_t *cursor, uint32_t *out_rows)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nq = (uint64_t)gridDim.x * 64;
    const int q = threadIdx.x & 3;
    for (uint64_t row = t >> 2; row < rows; row += nq) {
        const uint64_t b = row_ptr[row], e = row_ptr[row + 1];
        for (uint64_t k = b + q; k < e; k += 4) {
            const uint32_t s = src[k];
            out_rows[out_ptr[s] + atomicAdd(&cursor[s], 1u)] = (uint32_t)row;
        }
    }
}

// ---- unfused epilogue (edge-partition mode, after the all-reduce) ----------------------
// changed detection over ALL rows (every rank needs the full next frontier), estimator and
// Kahan only for the rows this rank owns.
__global__ __launch_bounds__(256) void epilogue_kernel(const PassParams p)
{
    __shared__ double s_raw[kTableLen];
    __shared__ double s_bias[kTableLen];
    __shared__ uint8_t s_lc[68];
    for (int i = threadIdx.x; i < kTableLen; i += 256) {
        s_raw[i] = p.raw[i];
        s_bias[i] = p.bias[i];
    }
    if (threadIdx.x < 65) s_lc[threadIdx.x] = p.lc[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 2, q = lane & 3, qshift = lane & ~3;
    const uint64_t ntiles = (p.row_hi - p.row_lo + 63) >> 6;
    unsigned long long cnt_changed = 0, cnt_out = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t row16 = p.row_lo + (tile << 6) + ((uint64_t)wave << 4);
        const uint64_t row = row16 + (uint64_t)g;
        const bool valid = row < p.row_hi;
        const uint32_t prev16 = (uint32_t)((const uint16_t *)p.bits_rd)[row16 >> 4];
        const uint32_t kd16 = (uint32_t)((const uint16_t *)p.kdirty)[row16 >> 4];
        const bool self_prev = (prev16 >> g) & 1u;
        const bool kd = (kd16 >> g) & 1u;
        // changed-only exchange: rows outside the union of the ranks' locally changed rows cannot have changed
        const uint32_t u16 = p.ubits ? (uint32_t)((const uint16_t *)p.ubits)[row16 >> 4] : 0xFFFFu;
        if (!(u16 | kd16)) { // wave-uniform: nothing to look at in these 16 rows
            if (lane == 0 && row16 < p.row_hi) ((uint16_t *)p.bits_wr)[row16 >> 4] = 0;
            continue;
        }
        uint4 oldv = make_uint4(0, 0, 0, 0), newv = oldv;
        if (valid && (((u16 >> g) & 1u) || !p.ubits)) {
            oldv = p.rd[row * 4 + q];
            newv = p.wr[row * 4 + q];
        }
        (void)self_prev;
        const uint64_t bal = __ballot(valid && u4_ne(oldv, newv));
        const bool changed = ((bal >> qshift) & 0xFull) != 0;
        const uint32_t ch16 = pack16(bal);
        if (lane == 0 && row16 < p.row_hi) ((uint16_t *)p.bits_wr)[row16 >> 4] = (uint16_t)ch16;
        cnt_changed += __popc(ch16);
        if (changed && q == 0) cnt_out += p.outdeg[row];
        bool err_nz = false;
        const bool mine = valid && row >= p.slice_lo && row < p.slice_hi;
        if (mine && (changed || kd)) {
            const uint64_t sz_old = p.size[row];
            const uint64_t sz_new = changed ? hll_size_quad(newv, s_raw, s_bias, s_lc) : sz_old;
            if (q == 0) {
                double ks = p.ksum[row], ke = p.kerr[row];
                err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1);
                if (err_nz) {
                    p.ksum[row] = ks;
                    p.kerr[row] = ke;
                }
                if (changed) p.size[row] = sz_new;
            }
        }
        const uint32_t nk16 = pack16(__ballot(err_nz));
        if (lane == 0 && row16 < p.row_hi && (nk16 | kd16)) ((uint16_t *)p.kdirty)[row16 >> 4] = (uint16_t)nk16;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt_out += __shfl_down(cnt_out, off);
    const unsigned long long v[4] = {cnt_changed, 0, 0, cnt_out};
    block_add_counters(p.counters, v, 0x9u);
}

// src_jp[e] for every entry of the work rows' source lists: the ONE register HyperLogLog::add(id) sets in the
// source's initial counter (same arithmetic as init_kernel below), as index | value << 8; virtual sources: 0.
__global__ __launch_bounds__(256) void src_jp_kernel(const uint32_t *src, uint64_t len, const uint64_t *id_low, const uint32_t *sid_of,
                                                     uint64_t n_pad, uint16_t *jp)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < len; e += (uint64_t)gridDim.x * 256) {
        const uint32_t s = src[e];
        uint16_t v = 0;
        if (s < n_pad && sid_of[s] != kNone) {
            const uint64_t hash = id_low[s] * 11400714819323198549ull;
            const uint32_t j = (uint32_t)(hash >> 58);
            const uint64_t w = hash << 6;
            const uint32_t pval = (w == 0 ? 64u : (uint32_t)__clzll((long long)w)) + 1u;
            v = (uint16_t)(j | (pval << 8));
        }
        jp[e] = v;
    }
}

// ---- initialisation: counter = HLL::default(); add_u128(id) (harmonic.rs:60-66) ----------
// HyperLogLog::add, hyperloglog.rs:4385-4396 with FastHasher (:4311-4313); only the low 64
// bits of the id are hashed (:4398-4400).
__global__ __launch_bounds__(256) void init_kernel(const uint64_t *id_low, const uint32_t *sid_of, uint64_t n_pad, uint4 *a,
                                                   uint4 *b, double *ksum, double *kerr, uint64_t *size,
                                                   uint32_t *bits, uint32_t *kdirty, const double *raw,
                                                   const double *bias, const uint8_t *lc)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = t >> 2;
    const int q = (int)(t & 3);
    if (row >= n_pad) return; // n_pad is a multiple of 64, so whole quads/waves exit together
    const bool real = sid_of[row] != kNone; // padding rows: all-zero counter, never changed
    uint4 v = make_uint4(0, 0, 0, 0);
    if (real) {
        const uint64_t hash = id_low[row] * 11400714819323198549ull;
        const uint32_t j = (uint32_t)(hash >> 58);
        const uint64_t w = hash << 6;
        const uint32_t pval = (w == 0 ? 64u : (uint32_t)__clzll((long long)w)) + 1u;
        if ((int)(j >> 4) == q) {
            const uint32_t word = (j & 15u) >> 2, byte = j & 3u;
            uint32_t ww[4] = {0, 0, 0, 0};
            ww[word] = pval << (8 * byte);
            v = make_uint4(ww[0], ww[1], ww[2], ww[3]);
        }
    }
    a[row * 4 + q] = v;
    (void)b; // `new = old.clone()` (harmonic.rs:67) needs no copy: pass 0 is always dense and writes every row of the other buffer
    // size() of a counter with exactly one register set: 63 zero registers -> the linear-counting branch
    // (hyperloglog.rs:4504-4515) -> lc[63]; the general estimator gives the same value by construction
    // (tests/test_gpu.py compares the cached sizes with the oracle's after hb_begin)
    (void)raw;
    (void)bias;
    const uint64_t sz = (uint64_t)lc[63];
    if (q == 0) {
        ksum[row] = 0.0;
        kerr[row] = 0.0;
        size[row] = real ? sz : 0;
    }
    // every node starts in the changed set (harmonic.rs:221-225): 16 rows per wave
    const uint32_t m16 = pack16(__ballot(real));
    if ((threadIdx.x & 63) == 0) {
        ((uint16_t *)bits)[row >> 4] = (uint16_t)m16;
        ((uint16_t *)kdirty)[row >> 4] = 0;
    }
}

// ---- helpers ---------------------------------------------------------------------------
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
// out[sid] for sid in ascending-NodeID order: f64::from(KahanSum) = sum (kahan_sum.rs:35-39);
// kept iff > 0.0, then / norm, non-finite -> 0.0; absent nodes are marked -1.0.
__global__ __launch_bounds__(256) void finish_kernel(const double *ksum, const uint32_t *dev_of, uint64_t n,
                                                     double norm, double *out, unsigned long long *count)
{
    unsigned long long kept = 0;
    for (uint64_t sid = (uint64_t)blockIdx.x * 256 + threadIdx.x; sid < n; sid += (uint64_t)gridDim.x * 256) {
        const double s = ksum[dev_of[sid]];
        double v = -1.0;
        if (s > 0.0) {
            v = s / norm;
            if (!(fabs(v) <= 1.7976931348623157e308)) v = 0.0; // is_finite
            kept++;
        }
        out[sid] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off);
    const unsigned long long v[4] = {kept, 0, 0, 0};
    block_add_counters(count, v, 0x1u); // striped: the host sums word 0 of every stripe
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
