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

// -DHB_DEBUG_BOUNDS (`make bounds` in this directory): device-side checks of every index the pass kernels gather through -
// source ids against the number of work rows, LDS list / strip positions against their capacity.  A failed check records
// its source line in a device word (first failure wins) and execution continues; the library reads the word at the end of
// every C-ABI call and fails that call with the line (hb_api.hip: guarded()).  No printf, no trap: both change what the
// kernel is (hostcall buffers, scratch) and a trapped queue loses the message.  The shipped build compiles them out.
#ifdef HB_DEBUG_BOUNDS
__device__ unsigned int g_dbg_line = 0;
#define HB_DBG_ASSERT(cond)                                                  \
    do {                                                                     \
        if (!(cond)) atomicCAS(&hbk::g_dbg_line, 0u, (unsigned int)__LINE__); \
    } while (0)
#else
#define HB_DBG_ASSERT(cond) ((void)0)
#endif

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
    uint64_t rows_total;         // n_pad + virtual rows: every source id is below it (checked at load: validate_plan_kernel)
    uint64_t slice_lo, slice_hi; // rows whose Kahan state this rank owns
    double t_plus_1;          // (t + 1) as f64, harmonic.rs:174
    // edge partition + HB_FLAG_CHANGED_ONLY: the unfused node-row launch records which rows its LOCAL merge changed
    // (lbits); after the union over the ranks only those rows are exchanged, and the epilogue visits only them (ubits)
    uint32_t *lbits;
    const uint32_t *ubits;
    // [r6] lean pass 0 (single rank, fused, INIT): hb_begin has NOT materialised the initial state - every node's counter is
    // HyperLogLog::default() + add(id) (harmonic.rs:60-62: one register, a function of id_low), its KahanSum is (0, 0) and its cached
    // size() is lc[63] - so pass 0 derives the own counters from id_low instead of reading 64 B per row that an init kernel would
    // first have had to write, takes the Kahan / size words as constants and stores them for EVERY row.  rd_init = the "old" buffer
    // itself: a row pass 0 leaves unchanged is stored there too, so that the lazy double buffer's invariant (the other buffer holds the
    // row's value unless it changed in this or the previous pass) holds for pass 1 in every pass mode.  NULL = the state is in memory.
    uint4 *rd_init;
    // [r6] pass 0 (INIT): one bit per WORK row, set = the row's sources are virtual rows.  The streaming form of pass 0 never looks at a
    // source id, and finding out a row's kind by loading its first one (`src[beg]`, as every other launch does - they read the list anyway)
    // pulled the whole index array through the memory system once more: 6.7 GB next to the 3.3 GB of src_jp the level-1 launch streams at
    // C4.  Written once at load (virt_rows_kernel).  NULL = read src[beg].
    const uint32_t *virt_rows;
    uint32_t xflags;          // experiments build only (0 in the product): bit 0 = the dense fused node rows neither read nor write size[] -
                              // WRONG RESULTS, a timing probe: what a node-row state diet of 16 B per row could buy at most
    const uint16_t *self_jp;  // per device row: register index | value << 8 of the node's OWN initial counter (0 = padding row), written at
                              // load time with the same arithmetic as src_jp (lean pass 0)
};

// the one register HyperLogLog::default(); add_u128(id) sets (hyperloglog.rs:4385-4400 with FastHasher :4311-4313; only the low 64 bits of
// the id are hashed), as index | value << 8 - the entry format of src_jp / self_jp
__device__ __forceinline__ uint16_t initial_register_jp(uint64_t id_low)
{
    const uint64_t hash = id_low * 11400714819323198549ull;
    const uint32_t j = (uint32_t)(hash >> 58);
    const uint64_t w = hash << 6;
    const uint32_t pval = (w == 0 ? 64u : (uint32_t)__clzll((long long)w)) + 1u;
    return (uint16_t)(j | (pval << 8));
}
// lane q's quarter (registers 16 q .. 16 q + 15) of the counter that entry describes
__device__ __forceinline__ uint4 counter_quarter_of_jp(uint32_t jp, int q)
{
    const uint32_t j = jp & 63u, pval = jp >> 8;
    uint32_t ww[4] = {0, 0, 0, 0};
    if (pval && (int)(j >> 4) == q) ww[(j & 15u) >> 2] = pval << (8 * (j & 3u));
    return make_uint4(ww[0], ww[1], ww[2], ww[3]);
}

} // namespace hbk
#include "hb_estimator.hip.h"
namespace hbk {

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
    const bool p_lean = INIT && REAL && FUSED && p.rd_init != nullptr;
    auto flush_pending = [&]() {
        bool err_nz = false;
        if (q < npend && (p_flags & 1u) && ((p_flags & 6u) || p_lean)) { // (lean pass 0: every row's words are written for the first time)
            const uint64_t sz_old = p_sz;
            uint64_t sz_new = sz_old;
            if (p_flags & 2u) sz_new = (p_flags & 8u) ? p_szfull : hll_size_from(p_sum, p_zeros, s_raw, s_bias, s_lc);
            double ks = p_ks, ke = p_ke;
            err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1); // still moving -> visit again
            if (err_nz || p_lean) {
                p.ksum[p_row] = ks;
                p.kerr[p_row] = ke;
            }
#ifdef HB_EXPERIMENTS
            if (!(p.xflags & 1u))
#endif
            if ((p_flags & 2u) || p_lean) p.size[p_row] = sz_new;
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
    uint32_t nvirt = 0; // (INIT) the row's sources are virtual rows (PassParams::virt_rows)
    uint4 nself = make_uint4(0, 0, 0, 0);
    const bool by_bitmap = INIT && p.virt_rows != nullptr;             // kernel-uniform
    const bool lean = INIT && REAL && FUSED && p.rd_init != nullptr; // kernel-uniform (PassParams::rd_init)
    auto own_counter = [&](uint64_t r) -> uint4 { // the row's counter before this pass
        if (INIT && REAL && FUSED && lean) return counter_quarter_of_jp((uint32_t)p.self_jp[r], q);
        return p.rd[r * 4 + q];
    };
    {
        const uint64_t r0 = row_lo + (tile0 << 6) + ((uint64_t)wave << 4) + (uint64_t)g;
        if (tile0 < ntiles && r0 < row_hi) {
            nbeg = p.row_ptr[r0];
            nend = p.row_ptr[r0 + 1];
            if (REAL && FUSED) nod = p.outdeg[r0];
            if (kDenseReal) nself = own_counter(r0);
            if (INIT && by_bitmap) nvirt = (p.virt_rows[r0 >> 5] >> (r0 & 31u)) & 1u;
        }
    }
    for (uint64_t tile = tile0; tile < ntiles; tile += tstride) {
        const uint64_t row16 = row_lo + (tile << 6) + ((uint64_t)wave << 4); // first row of this wave
        const uint64_t row = row16 + (uint64_t)g;
        const bool valid = row < row_hi;
        const uint64_t beg = nbeg, end = nend;
        const uint32_t od = nod; // out-degree of the row's node, used in the epilogue
        const uint32_t virt = nvirt;
        uint4 selfv = nself;
        {   // requests for the next tile of this workgroup
            const uint64_t nrow = row + (tstride << 6);
            nbeg = nend = 0;
            nod = 0;
            nvirt = 0;
            nself = make_uint4(0, 0, 0, 0);
            if (tile + tstride < ntiles && nrow < row_hi) {
                nbeg = p.row_ptr[nrow];
                nend = p.row_ptr[nrow + 1];
                if (REAL && FUSED) nod = p.outdeg[nrow];
                if (kDenseReal) nself = own_counter(nrow);
                if (INIT && by_bitmap) nvirt = (p.virt_rows[nrow >> 5] >> (nrow & 31u)) & 1u;
            }
        }
        // dense fused node rows: 4 of 5 rows change, so the estimator/Kahan words are requested now, unconditionally
        uint64_t pre_sz = 0;
        double pre_ks = 0.0, pre_ke = 0.0;
        if (kDenseReal && FUSED && valid) {
            if (lean) {
                // (0, 0) and size() of a counter with one register set = the linear-counting value for 63 zero registers
                // (hyperloglog.rs:4504-4515; init_kernel writes the same); a padding row's counter is empty and stays so: 0
                const bool real_row = ((__ballot(u4_ne(selfv, make_uint4(0, 0, 0, 0))) >> qshift) & 0xFull) != 0;
                pre_sz = real_row ? (uint64_t)s_lc[63] : 0ull;
            } else if (kEpi4) {
                if (q == npend) { // this lane owns the row's deferred epilogue
#ifdef HB_EXPERIMENTS
                    if (!(p.xflags & 1u))
#endif
                    pre_sz = p.size[row];
                    pre_ks = p.ksum[row];
                    pre_ke = p.kerr[row];
                }
            } else {
                pre_sz = p.size[row];
                if (q == 0) {
                    pre_ks = p.ksum[row];
                    pre_ke = p.kerr[row];
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
            uint32_t first;
            bool real_src;
            if (INIT && by_bitmap) { // pass 0: a row that streams its sources never touches the index array (PassParams::virt_rows)
                real_src = !virt;
                first = real_src ? 0u : p.src[beg];
                HB_DBG_ASSERT(real_src == (p.src[beg] < p.n_pad));
            } else {
                first = p.src[beg];
                real_src = first < p.n_pad;
            }
            const uint4 *base = real_src ? p.rd : (const uint4 *)(p.part - p.n_pad * 4);
            for (uint64_t e = beg; e < end; e += 4 * UNROLL) {
                uint32_t idx[UNROLL];
                if (INIT && real_src) {
                    // pass 0: the sources' single registers come with the edge list.  Each lane max-accumulates ITS sources
                    // into the row's 64 x u32 scratch counter in LDS (one ds_max_u32 per source) - nothing is broadcast
                    // to the other lanes of the quad and no lane tests registers that are not its own
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        const uint64_t ee = e + 4 * u + q;
                        idx[u] = (ee < end) ? (uint32_t)p.src_jp[ee] : 0u; // value 0 = nothing to merge
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
                    idx[u] = (ee < end) ? p.src[ee] : kNone;
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
                        HB_DBG_ASSERT(s0 < p.rows_total && s1 < p.rows_total && s2 < p.rows_total && s3 < p.rows_total);
                        HB_DBG_ASSERT((s0 < p.n_pad) == real_src && (s1 < p.n_pad) == real_src && (s2 < p.n_pad) == real_src && (s3 < p.n_pad) == real_src);
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
        const uint4 accv = acc_value(acc);
        const bool lane_diff = need && (!REAL || u4_ne(accv, selfv));
        const uint64_t bal = __ballot(lane_diff);
        const bool changed = ((bal >> qshift) & 0xFull) != 0;
        if (REAL) {
            // lazy double buffer: wr[row] already holds the right value unless the row changed in this or in the previous pass
            // [r5: used by the fused dense pass too - a fifth of the hosts of an R-MAT graph have no in-link at all and never
            // change: 64 B per such row and pass that nobody needs; the unfused forms store every row: the exchanges read them]
            if (need && (!FUSED || INIT || changed || self_prev || p.t_plus_1 == 1.0)) p.wr[row * 4 + q] = accv; // (pass 0 fills the other buffer)
            if (lean && need && !changed) p.rd_init[row * 4 + q] = accv; // (see PassParams::rd_init; a changed row is rewritten by pass 1 anyway)
        } else {
            if (changed) p.part[(row - p.n_pad) * 4 + q] = accv;
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
            hll_sum_quad(accv, rsum, zeros, big, s_lc);
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
            if (need && (changed || kd || lean)) {
                const uint64_t sz_old = kDenseReal ? pre_sz : p.size[row];
                const uint64_t sz_new = changed ? hll_size_quad(accv, s_raw, s_bias, s_lc) : sz_old;
                if (q == 0) {
                    double ks = kDenseReal ? pre_ks : p.ksum[row], ke = kDenseReal ? pre_ke : p.kerr[row];
                    err_nz = kahan_update(ks, ke, sz_new, sz_old, p.t_plus_1); // still moving -> visit again
                    if (err_nz || lean) {
                        p.ksum[row] = ks;
                        p.kerr[row] = ke;
                    }
                    if (changed || lean) p.size[row] = sz_new;
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
// together with the row's own counter: three round trips per row instead of up to fourteen.  COMPACT (default): every quad
// first packs the sources that passed the bit test to the front of an LDS strip, so the wave runs only as many gather rounds as
// its fullest quad needs; !COMPACT (tune[1] bit 13) walks the slots in place and skips a pair only if no quad of the wave uses it.
// W = index slots per lane and batch: 16 (64 sources per quad: hub chunks) or 4 (16 sources: node rows have ~5)
// Two filters in front of the bitmap test were built and measured SLOWER (DESIGN.md "tried and rejected", round 3): an
// LDS-staged coarse summary of the bitmap, and a "hot prefix" shortcut (sources below the first changed segment decided by
// <= 8 register compares): the bit tests they save are L1/L2 hits that overlap with the rest of the row.  A software pipeline over
// the tiles (row pointers / indices / bit words of the next tiles requested ahead) was slower too: it costs a wave per SIMD.
template <bool REAL, bool FUSED, bool STATS, int W, bool COMPACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void frontier_kernel(const PassParams p)
{
    __shared__ uint32_t s_cmp[COMPACT ? 64 * (4 * W + 1) : 1]; // per quad: its surviving source indices, packed
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
            nbeg = p.row_ptr[r];
            nend = p.row_ptr[r + 1];
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
                        idx[j] = (ee < end) ? p.src[ee] : kNone;
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
                    for (int j = 4 * b; j < 4 * b + 4; j++) {
                        HB_DBG_ASSERT(idx[j] == kNone || (idx[j] < p.rows_total && (idx[j] < p.n_pad) == real_src));
                        wb[j] = (idx[j] != kNone) ? p.bits_rd[idx[j] >> 5] : 0u;
                    }
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
            // ---- round trip 3: the gathers that are needed
            if (COMPACT) {
                // Few slots survive the bit test (C3 bitmap passes: ~10 % of a hub chunk's sources, ~1 of a node row's),
                // but which ones differs per quad, so a wave-uniform "skip this slot" rarely fires.  Each quad therefore
                // packs its surviving indices to the front of a 4W-entry LDS strip (prefix over the quad's 4 lanes by DPP;
                // strips are padded to an odd stride so 16 quads do not share a bank) and the wave runs only as many
                // gather rounds as its fullest quad needs.  The order of the sources changes; max does not care.
                uint32_t mine = 0;
#pragma unroll
                for (int j = 0; j < W; j++) mine += (idx[j] != kNone);
                uint32_t incl = mine;
                {
                    const uint32_t t1 = quad_perm<0x90>(incl); // lane q reads lane q-1
                    if (q >= 1) incl += t1;
                    const uint32_t t2 = quad_perm<0x44>(incl); // lane q reads lane q-2
                    if (q >= 2) incl += t2;
                }
                const uint32_t total = quad_bcast<3>(incl);
                uint32_t *strip = &s_cmp[(wave * 16 + g) * (4 * W + 1)];
                uint32_t pos = incl - mine;
#pragma unroll
                for (int j = 0; j < W; j++) {
                    if (idx[j] != kNone) {
                        HB_DBG_ASSERT(pos < (uint32_t)(4 * W));
                        strip[pos++] = idx[j];
                    }
                }
                // same wave writes and reads the strip: LDS operations of one wave complete in order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
                for (uint32_t j = 0; j < (uint32_t)(4 * W); j += 8) {
                    if (!__ballot(j < total)) break; // no quad of the wave has an entry left
                    uint4 r[2][4];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t e = j + 4 * u + q;
                        const uint32_t my = (e < total) ? strip[e] : kNone;
                        const uint32_t s0 = quad_bcast<0>(my), s1 = quad_bcast<1>(my);
                        const uint32_t s2 = quad_bcast<2>(my), s3 = quad_bcast<3>(my);
                        r[u][0] = r[u][1] = r[u][2] = r[u][3] = make_uint4(0, 0, 0, 0); // max with 0 = identity
                        if (s0 != kNone) r[u][0] = base[(uint64_t)s0 * 4 + q];
                        if (s1 != kNone) r[u][1] = base[(uint64_t)s1 * 4 + q];
                        if (s2 != kNone) r[u][2] = base[(uint64_t)s2 * 4 + q];
                        if (s3 != kNone) r[u][3] = base[(uint64_t)s3 * 4 + q];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) acc_merge(acc, r[0][k]);
                    if (__ballot(j + 4 < total)) {
#pragma unroll
                        for (int k = 0; k < 4; k++) acc_merge(acc, r[1][k]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the strip is rewritten by the next chunk / tile
            } else {
                // two slots (8 sources per quad) at a time, skipping a pair only if no quad of the wave uses it
#pragma unroll
                for (int j = 0; j < W; j += 2) {
                    if (!__ballot((idx[j] != kNone) | (idx[j + 1] != kNone))) continue;
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
            if (need && (changed || self_prev)) p.wr[row * 4 + q] = accv;
        } else {
            if (changed) p.part[(row - p.n_pad) * 4 + q] = accv;
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

} // namespace hbk
#include "hb_sweep.hip.h"
#ifdef HB_EXPERIMENTS
#include "hb_tail.hip.h" // the far tail as one workgroup: experiments build only (measured no faster)
#endif
namespace hbk {

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
// [r6] gathered from self_jp (2 bytes per node: 200 MB at C4, Infinity-Cache resident) instead of recomputed from id_low / sid_of
// (12 bytes per node behind two random 64-byte fetches per entry: 180 GB of HBM traffic for C4's 2.7 G entries, profiles/r05c_C4_pmc.json)
__global__ __launch_bounds__(256) void src_jp_kernel(const uint32_t *src, uint64_t len, const uint16_t *self_jp, uint64_t n_pad, uint16_t *jp)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < len; e += (uint64_t)gridDim.x * 256) {
        const uint32_t s = src[e];
        jp[e] = s < n_pad ? self_jp[s] : (uint16_t)0;
    }
}
// ---- pass 0, first level of hub chunks [r6] ------------------------------------------------------------------------------------------
// Every source of such a row is a real node whose counter still holds ONE register (harmonic.rs:60-62): the row streams its 2-byte
// src_jp entries, max-accumulates them in its scratch counter in LDS and writes the partial - what pass_kernel<false, .., INIT> does for
// these rows, as a kernel of its own: the generic kernel carries the gather path of the upper levels (16 uint4 in flight per lane: 92
// VGPRs, 5 waves per SIMD) through a launch that never gathers, and this launch answers to occupancy (a deeper per-wave pipeline was
// measured slower, profiles/r06g_*REJECTED*).  A quad takes all (<= 64) entries of its row at once, 16 per lane, quarters no row of the
// wave reaches are skipped wave-uniformly.  Same maxima, same bits (the per-pass parity variants run it: it is the default).
__global__ __launch_bounds__(256) void init_level1_kernel(const PassParams p)
{
    __shared__ uint4 s_cnt4[4 * 16 * 16]; // per wave 16 scratch counters of 64 x u32, register r of row g at word (r + 4 g) & 63 (pass_kernel's layout)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 2, q = lane & 3;
    uint32_t *cnt_row = (uint32_t *)s_cnt4 + (wave * 16 + g) * 64;
#pragma unroll
    for (int k = 0; k < 4; k++) s_cnt4[(wave * 16 + g) * 16 + ((4 * q + k + g) & 15)] = make_uint4(0, 0, 0, 0);
    const uint64_t ntiles = (p.row_hi - p.row_lo + 63) >> 6;
    constexpr int W = 16;
    uint64_t nbeg = 0, nend = 0;
    {
        const uint64_t r0 = p.row_lo + ((uint64_t)blockIdx.x << 6) + ((uint64_t)wave << 4) + (uint64_t)g;
        if (blockIdx.x < ntiles && r0 < p.row_hi) {
            nbeg = p.row_ptr[r0];
            nend = p.row_ptr[r0 + 1];
        }
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t row = p.row_lo + (tile << 6) + ((uint64_t)wave << 4) + (uint64_t)g;
        const bool valid = row < p.row_hi;
        const uint64_t beg = nbeg, end = nend;
        {
            const uint64_t nrow = row + ((uint64_t)gridDim.x << 6);
            nbeg = nend = 0;
            if (tile + gridDim.x < ntiles && nrow < p.row_hi) {
                nbeg = p.row_ptr[nrow];
                nend = p.row_ptr[nrow + 1];
            }
        }
        for (uint64_t e0 = beg; __ballot(e0 < end) != 0; e0 += 4 * W) { // (one iteration unless hb_options.chunk > 64)
            const uint64_t span = e0 < end ? end - e0 : 0;
            uint32_t jp[W];
#pragma unroll
            for (int b = 0; b < W / 4; b++) {
                if (__ballot(span > (uint64_t)(16 * b))) { // wave-uniform: some row of the wave reaches this quarter
#pragma unroll
                    for (int j = 4 * b; j < 4 * b + 4; j++) {
                        const uint64_t ee = e0 + 4 * j + q;
                        jp[j] = (ee < end && e0 < end) ? (uint32_t)p.src_jp[ee] : 0u; // value 0 = nothing to merge
                    }
                } else {
#pragma unroll
                    for (int j = 4 * b; j < 4 * b + 4; j++) jp[j] = 0u;
                }
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const uint32_t v = jp[j];
                if (v >> 8) __hip_atomic_fetch_max(&cnt_row[((v & 63u) + 4u * (uint32_t)g) & 63u], v >> 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // the rows' scratch counters -> register blocks (lane q: registers 16 q .. 16 q + 15), scratch cleared for the next tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t wv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint4 *cell = &s_cnt4[(wave * 16 + g) * 16 + ((4 * q + k + g) & 15)]; // registers 16 q + 4 k .. + 3
            const uint4 c = *cell;
            *cell = make_uint4(0, 0, 0, 0);
            wv[k] = c.x | (c.y << 8) | (c.z << 16) | (c.w << 24);
        }
        if (valid) p.part[(row - p.n_pad) * 4 + q] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// virt_rows (PassParams): bit r = work row r reads virtual rows; one lane per row, a wave writes the 64 bits of its rows
// *l0_virtual += the rows of [l0_lo, l0_hi) - the first level of hub chunks - that read virtual rows (none, by construction of both planners:
// init_level1_kernel, which streams every source of such a row, is only used when the count is 0)
__global__ __launch_bounds__(256) void virt_rows_kernel(const uint64_t *row_ptr, const uint32_t *src, uint64_t rows_total, uint64_t n_pad, uint32_t *bits,
                                                        uint64_t l0_lo, uint64_t l0_hi, unsigned long long *l0_virtual)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t r0 = (uint64_t)blockIdx.x * 256; r0 < rows_total; r0 += stride) { // wave-uniform trip count
        const uint64_t r = r0 + threadIdx.x;
        bool v = false;
        if (r < rows_total) {
            const uint64_t b = row_ptr[r], e = row_ptr[r + 1];
            v = b < e && src[b] >= n_pad;
        }
        const uint64_t m = __ballot(v);
        const uint64_t m0 = __ballot(v && r >= l0_lo && r < l0_hi);
        if (m0 && (threadIdx.x & 63) == 0) atomicAdd(l0_virtual, (unsigned long long)__popcll(m0));
        if ((threadIdx.x & 63) == 0 && r < rows_total) {
            bits[(r >> 5)] = (uint32_t)m;
            bits[(r >> 5) + 1] = (uint32_t)(m >> 32);
        }
    }
}
// self_jp[row]: the same entry for every node row's OWN initial counter (padding rows: 0); read by the lean pass 0 (PassParams::rd_init)
__global__ __launch_bounds__(256) void self_jp_kernel(const uint64_t *id_low, const uint32_t *sid_of, uint64_t n_pad, uint16_t *jp)
{
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_pad; r += (uint64_t)gridDim.x * 256)
        jp[r] = sid_of[r] != kNone ? initial_register_jp(id_low[r]) : (uint16_t)0;
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

// [r6] hb_begin when pass 0 runs lean (PassParams::rd_init): only the two bitmaps - every real node starts in the changed set
// (harmonic.rs:221-225), no Kahan state is dirty; counters, Kahan words and cached sizes are produced by pass 0 itself.
__global__ __launch_bounds__(256) void init_lean_kernel(const uint32_t *sid_of, uint64_t n_pad, uint32_t *bits, uint32_t *kdirty)
{
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x; // n_pad is a multiple of 64: whole waves
    if (row >= n_pad) return;
    const uint64_t bal = __ballot(sid_of[row] != kNone);
    if ((threadIdx.x & 63) == 0) {
        bits[row >> 5] = (uint32_t)bal;
        bits[(row >> 5) + 1] = (uint32_t)(bal >> 32);
        kdirty[row >> 5] = 0;
        kdirty[(row >> 5) + 1] = 0;
    }
}

} // namespace hbk
#include "hb_aux.hip.h"
