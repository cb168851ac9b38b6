// hb_sweep.hip.h - the data-driven (sweep) passes and the transposed work-row graph they run on.
// Part of the device code of stract_amd/csrc/hb_kernels.hip.h (included from there).
#pragma once

namespace hbk {

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
    // [r5] a pass queued BEFORE the host knows whether the previous one changed anything (hb_run's tail pipeline): the counter stripes
    // of the previous pass; all zero = that pass was the loop's last one (harmonic.rs:237-240) and every kernel of this one returns
    // at once, leaving the state exactly as that pass left it.  NULL = an ordinary pass.
    const unsigned long long *guard;
};

// wave-uniform: did the pass whose counters `guard` points at change any node (word 0 of its 64 stripes)?
__device__ __forceinline__ bool guard_open(const unsigned long long *guard)
{
    if (!guard) return true;
    const unsigned long long v = guard[4 * (threadIdx.x & (kStripes - 1))];
    return __ballot(v != 0ull) != 0ull;
}

__device__ __forceinline__ void touch_set(uint32_t *touch, uint32_t r, uint64_t rows_total)
{
    HB_DBG_ASSERT(r < rows_total);
    (void)rows_total;
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
    if (!guard_open(sp.guard)) return; // (a queued pass behind the loop's last one: EVERY kernel of it returns at once)
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
            HB_DBG_ASSERT(base < sp.p.n_pad);
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
    if (!guard_open(sp.guard)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t nseeds = sp.counts[0];
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (uint32_t i0 = wave * 64; i0 < nseeds; i0 += nwaves * 64) { // wave-uniform trip count
        const uint32_t i = i0 + lane;
        uint64_t b = 0, e = 0;
        uint32_t u = 0;
        if (i < nseeds) {
            u = sp.seeds[i];
            HB_DBG_ASSERT(u < sp.p.n_pad);
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
            if (item < total) touch_set(sp.touch, sp.out_rows[ob + (item - oex)], sp.p.rows_total);
        }
    }
}

// Convergence tail (a few thousand changed nodes at most): seed collection and expansion in ONE launch - every lane takes a
// word of the changed bitmap and walks the reader lists of its set bits itself; lists longer than 64 entries are walked by
// the whole wave (a hub that still changes this late is rare but must not serialise on one lane).  No seed list, no counts.
__global__ __launch_bounds__(256) void sweep_seed_small_kernel(const SweepParams sp)
{
    if (!guard_open(sp.guard)) return;
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
                for (uint64_t k = kb; k < ke; k++) touch_set(sp.touch, sp.out_rows[k], sp.p.rows_total);
        }
        uint64_t owners;
        while ((owners = __ballot(lng != 0)) != 0) {
            const int src = __ffsll((long long)owners) - 1;
            const uint32_t m = __shfl(lng, src);
            const int b = __ffs((int)m) - 1;
            if (lane == src) lng &= lng - 1;
            const uint64_t u = ((w0 + (uint64_t)(threadIdx.x & ~63) + (uint64_t)src) << 5) + (uint64_t)b;
            const uint64_t kb = sp.out_ptr[u], ke = sp.out_ptr[u + 1];
            for (uint64_t k = kb + lane; k < ke; k += 64) touch_set(sp.touch, sp.out_rows[k], sp.p.rows_total);
        }
    }
}

__global__ __launch_bounds__(256) void sweep_expand_heavy_kernel(const SweepParams sp)
{
    if (!guard_open(sp.guard)) return;
    const uint32_t nheavy = sp.counts[1];
    const uint64_t wbase = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64, nthreads = (uint64_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = 0; i < nheavy; i++) {
        const uint32_t u = sp.heavy[i];
        const uint64_t b = sp.out_ptr[u], e = sp.out_ptr[u + 1];
        for (uint64_t k0 = b + wbase; k0 < e; k0 += nthreads) { // wave-uniform trip count
            const uint64_t k = k0 + lane;
            if (k < e) touch_set(sp.touch, sp.out_rows[k], sp.p.rows_total);
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
template <bool REAL, bool SLOW = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void sweep_rows_kernel(const SweepParams sp)
{
    __shared__ double s_raw[REAL ? kTableLen : 1];
    __shared__ double s_bias[REAL ? kTableLen : 1];
    __shared__ uint8_t s_lc[68];
    __shared__ uint16_t s_list[4][2048]; // per wave: (owner lane << 5 | bit) of the set bits of its 64 words
    __shared__ uint32_t s_word[4][64];   // bitmap word index loaded by each lane
    __shared__ uint32_t s_chw[4][64];    // changed bits of this pass, per owned word
    __shared__ uint32_t s_kdw[4][64];    // Kahan-dirty bits, per owned word (REAL)
    // [r6] the sources of a touched row in THREE round trips, as the bitmap pass does it (frontier_kernel): all indices of the row at once
    // (W per lane: 64 sources for a hub chunk, 16 for a node row), then all their changed-bit words, then only the gathers that are needed,
    // packed per quad in an LDS strip so that the wave runs as many gather rounds as its fullest quad has survivors.  The loop this
    // replaces took 8 sources per round - index, bit word and gather each a dependent round trip - so a wave whose longest chunk row has
    // 64 sources walked 24 of them per batch of 16 rows while 1-2 sources per row had changed; at C4 the first sweep pass spent 3.5 ms in
    // the level-1 launch that way.  SLOW (experiments build, tune[1] bit 26): the old loop, kept as the A/B form.
    constexpr int W = REAL ? 4 : 16;
    __shared__ uint32_t s_strip[SLOW ? 1 : 64 * (4 * W + 1)];
    constexpr int kU = 2;                // (SLOW) index quads per gather round
    const PassParams &p = sp.p;
    if (!guard_open(sp.guard)) return; // (block-uniform: every wave reads the same words)
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
        // ([r5] the same rows as a streaming kernel of their own - 128 rows per wave step, 32 rows' loads in flight - was measured at C4,
        // where the first sweep pass revisits 35 M of them: node-row launch 3.67 -> 3.81 ms, and every later pass pays a scan of three
        // bitmaps: profiles/r05d_*; removed)
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
                HB_DBG_ASSERT(pos < 2048u);
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
            if (!SLOW) {
                for (uint64_t e0 = beg; e0 < end; e0 += 4 * W) { // one iteration unless the row has more than 4 W sources
                    // ---- round trip 1: all indices of the batch (slot j of lane q = source e0 + 4 j + q)
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
                    const uint32_t first = quad_bcast<0>(idx[0]);
                    const uint4 *srcbase = (first >= p.n_pad) ? (const uint4 *)(p.part - p.n_pad * 4) : p.rd;
                    // ---- round trip 2: the changed bits of all of them
                    uint32_t wb[W];
#pragma unroll
                    for (int b = 0; b < W / 4; b++) {
                        if (bal4[b]) {
#pragma unroll
                            for (int j = 4 * b; j < 4 * b + 4; j++) {
                                HB_DBG_ASSERT(idx[j] == kNone || idx[j] < p.rows_total);
                                wb[j] = (idx[j] != kNone) ? p.bits_rd[idx[j] >> 5] : 0u;
                            }
                        } else {
#pragma unroll
                            for (int j = 4 * b; j < 4 * b + 4; j++) wb[j] = 0u;
                        }
                    }
                    uint32_t mine = 0;
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        if (!((wb[j] >> (idx[j] & 31u)) & 1u)) idx[j] = kNone;
                        mine += (idx[j] != kNone);
                    }
                    lane_act |= mine != 0;
                    // ---- round trip 3: the survivors, packed per quad (prefix over the quad's 4 lanes by DPP; odd strip stride)
                    uint32_t incl = mine;
                    {
                        const uint32_t t1 = quad_perm<0x90>(incl); // lane q reads lane q-1
                        if (q >= 1) incl += t1;
                        const uint32_t t2 = quad_perm<0x44>(incl); // lane q reads lane q-2
                        if (q >= 2) incl += t2;
                    }
                    const uint32_t nsurv = quad_bcast<3>(incl);
                    uint32_t *strip = &s_strip[(wv * 16 + g) * (4 * W + 1)];
                    uint32_t pos = incl - mine;
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        if (idx[j] != kNone) {
                            HB_DBG_ASSERT(pos < (uint32_t)(4 * W));
                            strip[pos++] = idx[j];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // same wave writes and reads the strip: its LDS operations complete in order
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
                    for (uint32_t j = 0; j < (uint32_t)(4 * W); j += 8) {
                        if (!__ballot(j < nsurv)) break; // no quad of the wave has an entry left
                        uint4 r[2][4];
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const uint32_t e = j + 4 * u + q;
                            const uint32_t my = (e < nsurv) ? strip[e] : kNone;
                            const uint32_t s0 = quad_bcast<0>(my), s1 = quad_bcast<1>(my);
                            const uint32_t s2 = quad_bcast<2>(my), s3 = quad_bcast<3>(my);
                            r[u][0] = r[u][1] = r[u][2] = r[u][3] = make_uint4(0, 0, 0, 0); // max with 0 = identity
                            if (s0 != kNone) r[u][0] = srcbase[(uint64_t)s0 * 4 + q];
                            if (s1 != kNone) r[u][1] = srcbase[(uint64_t)s1 * 4 + q];
                            if (s2 != kNone) r[u][2] = srcbase[(uint64_t)s2 * 4 + q];
                            if (s3 != kNone) r[u][3] = srcbase[(uint64_t)s3 * 4 + q];
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) acc_merge(acc, r[0][k]);
                        if (__ballot(j + 4 < nsurv)) {
#pragma unroll
                            for (int k = 0; k < 4; k++) acc_merge(acc, r[1][k]);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the strip is rewritten by the next batch
                }
            } else if (beg < end) {
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
                    for (int u = 0; u < kU; u++) {
                        HB_DBG_ASSERT(idx[u] == kNone || idx[u] < p.rows_total);
                        wb[u] = (idx[u] != kNone) ? p.bits_rd[idx[u] >> 5] : 0u;
                    }
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
                    for (uint64_t k = sp.out_ptr[row]; k < sp.out_ptr[row + 1]; k++) touch_set(sp.touch, sp.out_rows[k], sp.p.rows_total);
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

} // namespace hbk
