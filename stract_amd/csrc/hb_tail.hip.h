// hb_tail.hip.h - the far convergence tail as ONE workgroup [r5]: whole passes over work LISTS, several passes per launch.
// Part of the device code of stract_amd/csrc/hb_kernels.hip.h (included from there).
//
// Why: once a pass changes a few thousand nodes at most, a sweep pass (hb_sweep.hip.h) is five 5-10 us launches that each scan
// a bitmap of the whole graph to find a handful of rows, and the tail of a deep graph is tens of such passes: 45 us per pass
// on the GPU, about as much again on the host to submit it (profiles/r05f_*: removing the host round trip alone bought nothing).
// Here the changed set is a LIST: one workgroup of 1024 lanes walks the seeds' reader lists (de-duplicated through the touch
// bitmap, whose bits it clears again), runs the touched rows level by level with __syncthreads() in between - the same row
// semantics as sweep_rows_kernel, bit for bit - keeps the changed / Kahan-dirty bitmaps exact BY LIST (the bits of two passes
// ago are cleared from their lists, nothing of size n is scanned or written), and goes on to the next pass by itself until a
// pass changes nothing (harmonic.rs:237-240), the changed set outgrows the lists, or the launch's pass budget is used up.
// Every bitmap is complete after every pass, so the multi-kernel path can take over at any pass boundary.
//
// Memory model: all communication is inside one workgroup on one CU (global stores + __syncthreads()); bitmap words that
// other lanes update with atomics in the same phase are read with L1-bypassing loads.
#pragma once

namespace hbk {

constexpr int kTailLevels = 6;             // virtual levels of a plan this kernel takes (deeper plans stay multi-kernel)
constexpr uint32_t kTailSeeds = 4096;      // a pass starts only if the previous one changed at most this many nodes ...
constexpr uint32_t kTailReaders = 1u << 14; // ... whose reader lists hold at most this many entries together
constexpr uint32_t kTailCap = 1u << 15;    // entries per list (seeds + readers + one parent per changed chunk stay far below)

enum : int { // words of TailParams::count
    kTcChanged = 0, // [0..1] node rows changed in pass t-1 / t-2 (by generation)
    kTcVirt = 2,    // [2..3] virtual rows changed in pass t-1 / t-2
    kTcDirty = 4,   // [4..5] Kahan-dirty node rows before / after the pass
    kTcPasses = 8,  // passes this launch completed
    kTcReason = 9,  // 0 = the loop ended (a pass changed nothing), 1 = pass budget used up, 2 = the next pass does not fit the lists
    kTcGen = 10,    // generation index the lists are in (which half holds "t-1")
    kTcWords = 16
};

struct TailParams {
    const uint64_t *row_ptr;
    const uint32_t *src;
    uint4 *regs[2];
    uint4 *part;
    uint32_t *bits[2];
    uint32_t *kdirty;
    double *ksum, *kerr;
    uint64_t *size;
    const uint32_t *outdeg;
    unsigned long long *counters; // base of the per-pass counter slots (kCounterWords each)
    const double *raw, *bias;
    const uint8_t *lc;
    const uint64_t *out_ptr;
    const uint32_t *out_rows;
    uint32_t *touch;
    uint64_t n_pad, rows_total;
    uint64_t level_begin[kTailLevels + 1];
    int levels;
    uint32_t *changed[2], *vchanged[2], *dirty[2];
    uint32_t *work; // (kTailLevels + 1) x kTailCap: touched rows per level, node rows last
    uint32_t *count;
    uint64_t t0;
    int cur0;
    uint32_t max_passes; // of this launch (hb_step: 1)
};

// ---- entry: the lists from the bitmaps (grid-wide, once per entry into the tail kernel) -------------------------------------
// generation 0: changed[0] = nodes changed in the previous pass (seeds), vchanged[0] = the virtual rows' bits in the bitmap the
// NEXT pass reads, dirty[0] = the Kahan-dirty nodes.  The stale bits of two passes ago - node part of the bitmap this pass
// writes, virtual part of the one it reads - are CLEARED here instead of being listed (at BASELINE sizes they are the changed set
// of a pass that still changed a million nodes: no list holds them; every kernel of the other path rewrites those parts before it
// reads them, so a cleared bitmap is as good to it as a stale one).  Counts keep running past the capacity: the loop kernel starts
// a pass only if everything fits.
__device__ __forceinline__ void tail_append_bits(uint32_t word, uint64_t first_row, uint32_t *list, uint32_t *count)
{
    const int lane = threadIdx.x & 63;
    const uint32_t mine = __popc(word);
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(incl, off);
        if (lane >= off) incl += a;
    }
    const uint32_t total = __shfl(incl, 63);
    if (!total) return; // wave-uniform
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(count, total);
    base = __shfl(base, 63) + incl - mine;
    while (word) {
        const int b = __ffs((int)word) - 1;
        word &= word - 1;
        if (base < kTailCap) list[base] = (uint32_t)(first_row + (uint64_t)b);
        base++;
    }
}
__global__ __launch_bounds__(256) void tail_collect_kernel(const TailParams P)
{
    const int cur = P.cur0;
    const uint64_t node_words = P.n_pad >> 5, all_words = (P.rows_total + 31) >> 5;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w0 = (uint64_t)blockIdx.x * 256; w0 < all_words; w0 += stride) { // wave-uniform trip count
        const uint64_t w = w0 + threadIdx.x;
        const bool in = w < all_words, node = w < node_words;
        const uint32_t a = in ? P.bits[cur][w] : 0u, b = in ? P.bits[cur ^ 1][w] : 0u, d = (in && node) ? P.kdirty[w] : 0u;
        tail_append_bits(node ? a : 0u, w << 5, P.changed[0], &P.count[kTcChanged + 0]);
        tail_append_bits(node ? 0u : b, w << 5, P.vchanged[0], &P.count[kTcVirt + 0]);
        tail_append_bits(d, w << 5, P.dirty[0], &P.count[kTcDirty + 0]);
        if (in && node && b) P.bits[cur ^ 1][w] = 0;  // changed two passes ago: this pass writes its own set there
        if (in && !node && a) P.bits[cur][w] = 0;     // virtual rows changed two passes ago: this pass sets its own
    }
}

// ---- quad-level helpers (rows are run one per quad; quads of a wave take different paths) ---------------------------------
__device__ __forceinline__ bool quad_any(bool v)
{
    uint32_t x = v ? 1u : 0u;
    x |= quad_perm<0xB1>(x);
    x |= quad_perm<0x4E>(x);
    return x != 0;
}
__device__ __forceinline__ bool tail_bit(const uint32_t *bits, uint64_t r)
{
    return (__hip_atomic_load(&bits[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (r & 31u)) & 1u;
}
// a row read by a changed source: first toucher appends it to the work list of its level
__device__ __forceinline__ void tail_touch(const TailParams &P, uint32_t r, uint32_t *s_work)
{
    const uint32_t bit = 1u << (r & 31u);
    if (__hip_atomic_load(&P.touch[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return;
    if (atomicOr(&P.touch[r >> 5], bit) & bit) return;
    int l = P.levels; // node rows last
    if (r >= P.n_pad) {
        l = 0;
        while (l + 1 < P.levels && r >= P.level_begin[l + 1]) l++;
    }
    const uint32_t k = atomicAdd(&s_work[l], 1u); // LDS
    if (k < kTailCap) P.work[(size_t)l * kTailCap + k] = r;
}
// gather of the row's sources whose changed bit is set (frontier semantics of sweep_rows_kernel); all 4 lanes of the quad
__device__ __forceinline__ bool tail_gather(const TailParams &P, const uint4 *rd, const uint32_t *bits_rd, uint64_t beg, uint64_t end, int q, Acc &acc)
{
    bool act = false;
    if (beg < end) {
        const uint32_t first = P.src[beg];
        const uint4 *base = (first >= P.n_pad) ? (const uint4 *)(P.part - P.n_pad * 4) : rd;
        for (uint64_t e = beg; e < end; e += 4) { // quad-uniform trip count
            const uint64_t ee = e + (uint64_t)q;
            uint32_t idx = (ee < end) ? P.src[ee] : kNone;
            if (idx != kNone && !tail_bit(bits_rd, idx)) idx = kNone;
            act |= idx != kNone;
            const uint32_t s0 = quad_bcast<0>(idx), s1 = quad_bcast<1>(idx), s2 = quad_bcast<2>(idx), s3 = quad_bcast<3>(idx);
            if (s0 != kNone) acc_merge(acc, base[(uint64_t)s0 * 4 + q]);
            if (s1 != kNone) acc_merge(acc, base[(uint64_t)s1 * 4 + q]);
            if (s2 != kNone) acc_merge(acc, base[(uint64_t)s2 * 4 + q]);
            if (s3 != kNone) acc_merge(acc, base[(uint64_t)s3 * 4 + q]);
        }
    }
    return quad_any(act);
}

// ---- the loop ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tail_loop_kernel(const TailParams P)
{
    __shared__ double s_raw[kTableLen];
    __shared__ double s_bias[kTableLen];
    __shared__ uint8_t s_lc[68];
    __shared__ uint32_t s_work[kTailLevels + 1];  // entries of the work lists of this pass
    __shared__ uint32_t s_n[6];                   // list lengths, indexed like P.count[0..5]
    __shared__ uint32_t s_readers, s_changed, s_touched, s_newdirty, s_newvirt;
    __shared__ unsigned long long s_out;
    for (int i = threadIdx.x; i < kTableLen; i += 1024) {
        s_raw[i] = P.raw[i];
        s_bias[i] = P.bias[i];
    }
    if (threadIdx.x < 65) s_lc[threadIdx.x] = P.lc[threadIdx.x];
    if (threadIdx.x < 6) s_n[threadIdx.x] = P.count[threadIdx.x];
    __syncthreads();
    const int q = threadIdx.x & 3;
    const uint32_t quad = threadIdx.x >> 2, nquads = 256;
    int g = (int)P.count[kTcGen] & 1; // lists of generation g hold "t-1", g ^ 1 hold "t-2" (and receive this pass' sets)
    uint32_t done = 0, reason = 1;
    for (; done < P.max_passes; done++) {
        const uint64_t t = P.t0 + done;
        const int cur = P.cur0 ^ (int)(done & 1u);
        uint4 *rd = P.regs[cur], *wr = P.regs[cur ^ 1];
        uint32_t *bits_rd = P.bits[cur], *bits_wr = P.bits[cur ^ 1];
        const uint32_t *seeds = P.changed[g];
        uint32_t *ch_new = P.changed[g ^ 1], *vc_new = P.vchanged[g ^ 1];
        const uint32_t *d_old = P.dirty[g];
        uint32_t *d_new = P.dirty[g ^ 1];
        const uint32_t n_seed = s_n[kTcChanged + g], n_old2 = s_n[kTcChanged + (g ^ 1)], n_vold2 = s_n[kTcVirt + (g ^ 1)], n_dold = s_n[kTcDirty + g];
        const double t_plus_1 = (double)(t + 1);
        // ---- does the pass fit?  (nothing has been modified yet: the multi-kernel path can run this pass instead)
        if (threadIdx.x == 0) s_readers = 0;
        __syncthreads();
        {
            uint32_t mine = 0;
            for (uint32_t i = threadIdx.x; i < n_seed && i < kTailCap; i += 1024) {
                const uint32_t u = seeds[i];
                mine += (uint32_t)min((uint64_t)kTailReaders + 1, P.out_ptr[u + 1] - P.out_ptr[u]);
            }
            if (mine) atomicAdd(&s_readers, mine);
        }
        __syncthreads();
        if (n_seed > kTailSeeds || n_old2 > kTailCap || n_vold2 > kTailCap || n_dold > kTailCap || s_readers > kTailReaders) {
            reason = 2;
            break; // (block-uniform)
        }
        if (threadIdx.x <= kTailLevels) s_work[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            s_changed = s_touched = s_newdirty = s_newvirt = 0;
            s_out = 0;
        }
        // ---- the bits of two passes ago leave the bitmaps this pass writes
        for (uint32_t i = threadIdx.x; i < n_old2; i += 1024) atomicAnd(&bits_wr[ch_new[i] >> 5], ~(1u << (ch_new[i] & 31u)));
        for (uint32_t i = threadIdx.x; i < n_vold2; i += 1024) atomicAnd(&bits_rd[vc_new[i] >> 5], ~(1u << (vc_new[i] & 31u)));
        __syncthreads();
        // ---- seeds -> touched rows (one lane per seed; reader lists are short here: kTailReaders in all)
        for (uint32_t i = threadIdx.x; i < n_seed; i += 1024) {
            const uint32_t u = seeds[i];
            for (uint64_t k = P.out_ptr[u]; k < P.out_ptr[u + 1]; k++) tail_touch(P, P.out_rows[k], s_work);
        }
        __syncthreads();
        // ---- virtual levels, bottom up: a changed chunk touches its readers (a higher level, or a node row)
        for (int l = 0; l < P.levels; l++) {
            const uint32_t nl = min(s_work[l], kTailCap);
            for (uint32_t i = quad; i < nl; i += nquads) { // quad-uniform
                const uint64_t row = P.work[(size_t)l * kTailCap + i];
                const uint64_t beg = P.row_ptr[row], end = P.row_ptr[row + 1];
                const uint4 selfv = P.part[(row - P.n_pad) * 4 + q];
                Acc acc;
                acc_zero(acc);
                (void)tail_gather(P, rd, bits_rd, beg, end, q, acc);
                acc_merge(acc, selfv);
                const uint4 accv = acc_value(acc);
                if (quad_any(u4_ne(accv, selfv))) {
                    P.part[(row - P.n_pad) * 4 + q] = accv;
                    if (q == 0) {
                        atomicOr(&bits_rd[row >> 5], 1u << (row & 31u)); // this pass' changed bit of the virtual row
                        const uint32_t k = atomicAdd(&s_newvirt, 1u);
                        if (k < kTailCap) vc_new[k] = (uint32_t)row;
                        for (uint64_t k2 = P.out_ptr[row]; k2 < P.out_ptr[row + 1]; k2++) tail_touch(P, P.out_rows[k2], s_work);
                    }
                }
            }
            __syncthreads();
        }
        // ---- touched node rows: merge, changed bit, estimator, Kahan (sweep_rows_kernel<true>'s row, one quad each)
        {
            const uint32_t nn = min(s_work[P.levels], kTailCap);
            for (uint32_t i = quad; i < nn; i += nquads) {
                const uint64_t row = P.work[(size_t)P.levels * kTailCap + i];
                const uint64_t beg = P.row_ptr[row], end = P.row_ptr[row + 1];
                const uint4 selfv = rd[row * 4 + q];
                Acc acc;
                acc_zero(acc);
                const bool touched = tail_gather(P, rd, bits_rd, beg, end, q, acc);
                acc_merge(acc, selfv);
                const uint4 accv = acc_value(acc);
                const bool changed = quad_any(u4_ne(accv, selfv));
                const bool self_prev = tail_bit(bits_rd, row), kd = tail_bit(P.kdirty, row);
                if (changed || self_prev) wr[row * 4 + q] = accv; // lazy double buffer
                if (changed || kd) {
                    const uint64_t sz_old = P.size[row];
                    const uint64_t sz_new = changed ? hll_size_quad(accv, s_raw, s_bias, s_lc) : sz_old;
                    if (q == 0) {
                        double ks = P.ksum[row], ke = P.kerr[row];
                        const bool err_nz = kahan_update(ks, ke, sz_new, sz_old, t_plus_1);
                        if (err_nz) {
                            P.ksum[row] = ks;
                            P.kerr[row] = ke;
                            const uint32_t k = atomicAdd(&s_newdirty, 1u);
                            if (k < kTailCap) d_new[k] = (uint32_t)row;
                        }
                        if (changed) P.size[row] = sz_new;
                        if (err_nz && !kd) atomicOr(&P.kdirty[row >> 5], 1u << (row & 31u));
                        if (!err_nz && kd) atomicAnd(&P.kdirty[row >> 5], ~(1u << (row & 31u)));
                    }
                }
                if (q == 0) {
                    if (touched) atomicAdd(&s_touched, 1u);
                    if (changed) {
                        atomicOr(&bits_wr[row >> 5], 1u << (row & 31u));
                        const uint32_t k = atomicAdd(&s_changed, 1u);
                        if (k < kTailCap) ch_new[k] = (uint32_t)row;
                        atomicAdd(&s_out, (unsigned long long)P.outdeg[row]);
                    }
                }
            }
        }
        // ---- rows no changed source reached: carry-over of last pass' changers, `+= 0.0` on the Kahan-dirty ones
        // (a row is in one of three places: touched (above), a seed, or dirty-only)
        for (uint32_t i = quad; i < n_seed; i += nquads) {
            const uint64_t row = seeds[i];
            if (tail_bit(P.touch, row)) continue; // quad-uniform
            wr[row * 4 + q] = rd[row * 4 + q];
        }
        for (uint32_t i = threadIdx.x; i < n_seed + n_dold; i += 1024) {
            const uint64_t row = i < n_seed ? seeds[i] : d_old[i - n_seed];
            if (tail_bit(P.touch, row)) continue;
            if (i >= n_seed && tail_bit(bits_rd, row)) continue; // also a seed: taken there
            if (!tail_bit(P.kdirty, row)) continue;
            double ks = P.ksum[row], ke = P.kerr[row];
            if (kahan_update(ks, ke, 0, 0, t_plus_1)) { // update_centralities with size(new) == size(old), harmonic.rs:159-176
                P.ksum[row] = ks;
                P.kerr[row] = ke;
                const uint32_t k = atomicAdd(&s_newdirty, 1u);
                if (k < kTailCap) d_new[k] = (uint32_t)row;
            } else {
                atomicAnd(&P.kdirty[row >> 5], ~(1u << (row & 31u)));
            }
        }
        __syncthreads();
        // ---- the touch bitmap is all zero again; the pass' counters; the lists change roles
        for (int l = 0; l <= P.levels; l++) {
            const uint32_t nl = min(s_work[l], kTailCap);
            for (uint32_t i = threadIdx.x; i < nl; i += 1024) {
                const uint32_t r = P.work[(size_t)l * kTailCap + i];
                atomicAnd(&P.touch[r >> 5], ~(1u << (r & 31u)));
            }
        }
        __syncthreads();
        const uint32_t changed_now = s_changed;
        if (threadIdx.x == 0) {
            unsigned long long *c = P.counters + (size_t)kCounterWords * t;
            c[0] = changed_now;
            c[2] = s_touched;
            c[3] = s_out;
            s_n[kTcChanged + (g ^ 1)] = changed_now;
            s_n[kTcVirt + (g ^ 1)] = s_newvirt;
            s_n[kTcDirty + (g ^ 1)] = s_newdirty;
        }
        __syncthreads();
        g ^= 1;
        if (changed_now == 0) {
            done++;
            reason = 0;
            break;
        }
    }
    if (threadIdx.x < 6) P.count[threadIdx.x] = s_n[threadIdx.x];
    if (threadIdx.x == 0) {
        P.count[kTcPasses] = done;
        P.count[kTcReason] = reason;
        P.count[kTcGen] = (uint32_t)g;
    }
}

} // namespace hbk
