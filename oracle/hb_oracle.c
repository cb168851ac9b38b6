/*
 * hb_oracle.c - CPU restatement of Stract's HyperBall harmonic centrality.
 *
 * TEST INFRASTRUCTURE ONLY (see hb_oracle.h).  PARITY STATUS: "parity unpinned" -
 * the Rust reference cannot be built here and its tests hold no numeric centrality
 * vectors; this file follows the reference line by line and is checked against the
 * reference's behavioural tests and its one exact-float known answer.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp; never -ffast-math).
 */
#include "hb_oracle.h"
#include "hll64_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------- */
/* HyperLogLog<64, FastHasher>                                                       */
/* ------------------------------------------------------------------------------- */

#define HLL_N 64
#define HLL_B 6 /* hyperloglog.rs:4381-4383: (64 as f64).log2() as usize */

/* hyperloglog.rs:4311-4313 FastHasher::hash, :4385-4396 add */
void hbo_hll_add(uint8_t reg[64], uint64_t item)
{
    uint64_t hash = item * 11400714819323198549ull; /* wrapping_mul */
    uint64_t j = hash >> (64 - HLL_B);
    uint64_t w = hash << HLL_B;
    /* p = leading_zeros(w) + 1; u64::leading_zeros(0) == 64 */
    unsigned lz = (w == 0) ? 64u : (unsigned)__builtin_clzll(w);
    uint8_t p = (uint8_t)(lz + 1);
    if (p > reg[j]) reg[j] = p;
}

/* hyperloglog.rs:4531-4535 */
void hbo_hll_merge(uint8_t dst[64], const uint8_t src[64])
{
    for (int i = 0; i < HLL_N; i++)
        if (src[i] > dst[i]) dst[i] = src[i];
}

/* ONE_OVER_POWER_OF_TWO[val] (hyperloglog.rs:4043-4300) == 2^-val exactly for
 * val in 0..255 (verified by oracle/gen_tables.py). */
static inline double pow2_neg(unsigned val)
{
    uint64_t bits = (uint64_t)(1023u - val) << 52;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

/* slice::binary_search_by(|v| v.total_cmp(&e)) as called at hyperloglog.rs:4413.
 * Returns the index i of Ok(i) / Err(i). */
static int bsearch_rust_1_82(const double *a, int len, double e)
{
    /* library/core/src/slice/mod.rs (Rust 1.82+): branchless halving that keeps `base` */
    int size = len, base = 0;
    while (size > 1) {
        int half = size / 2;
        int mid = base + half;
        /* cmp = a[mid].total_cmp(e); base = if cmp == Greater { base } else { mid } */
        if (!(a[mid] > e)) base = mid;
        size -= half;
    }
    if (a[base] == e) return base;          /* Ok(base)  */
    return base + (a[base] < e ? 1 : 0);    /* Err(base + (cmp == Less)) */
}

static int bsearch_classic(const double *a, int len, double e)
{
    /* Rust 1.52 .. 1.81 */
    int size = len, left = 0, right = len;
    while (left < right) {
        int mid = left + size / 2;
        if (a[mid] < e) left = mid + 1;
        else if (a[mid] > e) right = mid;
        else return mid; /* Ok(mid) */
        size = right - left;
    }
    return left; /* Err(left) */
}

int hbo_hll_bias_first_index(double e, int variant)
{
    const int len = HLL64_TABLE_LEN;
    int i = (variant == HBO_BSEARCH_CLASSIC) ? bsearch_classic(HLL64_RAW_ESTIMATE, len, e)
                                             : bsearch_rust_1_82(HLL64_RAW_ESTIMATE, len, e);
    /* hyperloglog.rs:4413-4416: Err(len) -> len-1, Ok(i) | Err(i) -> i */
    if (i == len) i = len - 1;
    return i;
}

/* hyperloglog.rs:4407-4470 estimate_bias(e, b) with b = 6 -> table index 1 */
double hbo_hll_estimate_bias(double e, int variant)
{
    const int len = HLL64_TABLE_LEN;
    const double *raw = HLL64_RAW_ESTIMATE;
    int idx_left = hbo_hll_bias_first_index(e, variant);         /* always Some */
    int idx_right = (idx_left < len - 1) ? idx_left + 1 : -1;     /* -1 == None   */
    int neighbors[6];
    for (int k = 0; k < 6; k++) { /* K = 6, :4408 */
        int right_instead_left, idx;
        if (idx_left >= 0 && idx_right >= 0) {
            double delta_left = fabs(raw[idx_left] - e);
            double delta_right = fabs(raw[idx_right] - e);
            if (delta_right < delta_left) { right_instead_left = 1; idx = idx_right; }
            else { right_instead_left = 0; idx = idx_left; }
        } else if (idx_left >= 0) {
            right_instead_left = 0; idx = idx_left;
        } else {
            /* (None, Some) - (None, None) is unreachable for len >= K */
            right_instead_left = 1; idx = idx_right;
        }
        neighbors[k] = idx;
        if (right_instead_left) idx_right = (idx < len - 1) ? idx + 1 : -1;
        else idx_left = (idx > 0) ? idx - 1 : -1;
    }
    /* :4469 neighbors.iter().map(|&i| bias_data[i]).sum::<f64>() / 6.0 (left fold) */
    double s = 0.0;
    for (int k = 0; k < 6; k++) s += HLL64_BIAS[neighbors[k]];
    return s / 6.0;
}

/* Rust `f64 as usize`: truncates toward zero, saturates, NaN -> 0 */
static inline uint64_t f64_as_usize(double x)
{
    if (!(x > 0.0)) return 0; /* negatives, -0, +0, NaN */
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}

uint64_t hbo_hll_size_ex(const uint8_t reg[64], int variant, double *e_out, double *e_star_out)
{
    const double m = 64.0; /* :4485 */
    /* :4488-4492 left fold in register order */
    double sum = 0.0;
    for (int i = 0; i < HLL_N; i++) sum += pow2_neg(reg[i]);
    double z = 1.0 / sum;                 /* :4494 */
    /* :4496 self.am() * m.powi(2) * z ; am() = 0.709 for 64 <= m < 128 (:4371-4372) */
    double e = 0.709 * (m * m) * z;
    double e_star = (e <= 5.0 * m) ? e - hbo_hll_estimate_bias(e, variant) : e; /* :4498-4502 */
    /* :4504 bytecount::count(&registers, 0) */
    unsigned v = 0;
    for (int i = 0; i < HLL_N; i++) v += (reg[i] == 0);
    /* :4505-4509, linear_counting :4472-4476  m * (m / v).ln() */
    double h = (v != 0) ? m * log(m / (double)v) : e_star;
    if (e_out) *e_out = e;
    if (e_star_out) *e_star_out = e_star;
    /* :4511-4515, threshold(6) = THRESHOLD_DATA_VEC[6-4] = 40 (:31,:4478-4480) */
    if (h <= 40.0) return f64_as_usize(h);
    return f64_as_usize(e_star);
}

uint64_t hbo_hll_size(const uint8_t reg[64])
{
    return hbo_hll_size_ex(reg, HBO_BSEARCH_RUST_1_82, NULL, NULL);
}

/* ------------------------------------------------------------------------------- */
/* KahanSum (kahan_sum.rs:47-54)                                                     */
/* ------------------------------------------------------------------------------- */
void hbo_kahan_add(double *sum, double *err, double rhs)
{
    double y = rhs - *err;
    double t = *sum + y;
    *err = (t - *sum) - y;
    *sum = t;
}

/* ------------------------------------------------------------------------------- */
/* dense HyperBall                                                                    */
/* ------------------------------------------------------------------------------- */
struct hbo_dense {
    uint64_t n;
    const uint64_t *row_ptr;
    const uint32_t *src;
    uint8_t *old_regs, *new_regs; /* n*64 each */
    uint8_t *changed_prev, *changed_next;
    double *ksum, *kerr;
    uint64_t *size_old;
    uint64_t t;
    int has_changes;
    int threads;
    int bsearch;
    uint64_t last_active, last_touched;
};

/* OpenMP threads worth starting for `work` units (edges + 8 * nodes): small problems on a many-core box
 * are slower with every hardware thread than with a few (fork/join and idle spinning dominate). */
static int eff_threads(int requested, uint64_t work)
{
#ifdef _OPENMP
    int nt = requested > 0 ? requested : omp_get_max_threads();
    uint64_t cap = work / 32768 + 1;
    if ((uint64_t)nt > cap) nt = (int)cap;
    return nt < 1 ? 1 : nt;
#else
    (void)requested; (void)work;
    return 1;
#endif
}

hbo_dense *hbo_dense_create(uint64_t n, const uint64_t *id_low64, const uint64_t *row_ptr,
                            const uint32_t *src, int threads)
{
    hbo_dense *s = (hbo_dense *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->n = n;
    s->row_ptr = row_ptr;
    s->src = src;
    s->threads = threads;
    s->bsearch = HBO_BSEARCH_RUST_1_82;
    size_t nn = n ? n : 1;
    /* malloc + parallel first touch below: on a multi-socket host the pages of the big arrays
     * end up spread over the NUMA nodes instead of all on the creating thread's node */
    s->old_regs = (uint8_t *)malloc(nn * 64);
    s->new_regs = (uint8_t *)malloc(nn * 64);
    s->changed_prev = (uint8_t *)malloc(nn);
    s->changed_next = (uint8_t *)malloc(nn);
    s->ksum = (double *)malloc(nn * sizeof(double));
    s->kerr = (double *)malloc(nn * sizeof(double));
    s->size_old = (uint64_t *)malloc(nn * sizeof(uint64_t));
    if (!s->old_regs || !s->new_regs || !s->changed_prev || !s->changed_next || !s->ksum ||
        !s->kerr || !s->size_old) {
        hbo_dense_destroy(s);
        return NULL;
    }
    /* initialize (harmonic.rs:53-73): counter = HLL::default(); add_u128(id); new = old.clone();
     * harmonic.rs:221-225: every node starts in the changed set */
    if (n == 0) {
        memset(s->old_regs, 0, 64);
        memset(s->new_regs, 0, 64);
        s->changed_prev[0] = 1; s->changed_next[0] = 0;
        s->ksum[0] = s->kerr[0] = 0.0; s->size_old[0] = 0;
    }
#ifdef _OPENMP
    int nt0 = eff_threads(threads, 8 * n);
#pragma omp parallel for schedule(static) num_threads(nt0)
#endif
    for (int64_t vi = 0; vi < (int64_t)n; vi++) {
        const uint64_t v = (uint64_t)vi;
        memset(s->old_regs + 64 * v, 0, 64);
        hbo_hll_add(s->old_regs + 64 * v, id_low64[v]);
        memcpy(s->new_regs + 64 * v, s->old_regs + 64 * v, 64);
        s->size_old[v] = hbo_hll_size_ex(s->old_regs + 64 * v, s->bsearch, NULL, NULL);
        s->changed_prev[v] = 1;
        s->changed_next[v] = 0;
        s->ksum[v] = 0.0;
        s->kerr[v] = 0.0;
    }
    s->has_changes = 1; /* harmonic.rs:232 */
    return s;
}

void hbo_dense_set_bsearch(hbo_dense *s, int variant)
{
    s->bsearch = variant;
    for (uint64_t v = 0; v < s->n; v++)
        s->size_old[v] = hbo_hll_size_ex(s->old_regs + 64 * v, s->bsearch, NULL, NULL);
}

void hbo_dense_destroy(hbo_dense *s)
{
    if (!s) return;
    free(s->old_regs); free(s->new_regs); free(s->changed_prev); free(s->changed_next);
    free(s->ksum); free(s->kerr); free(s->size_old);
    free(s);
}

/* First half of a pass: update_all_counters (harmonic.rs:116-157) over the edges this
 * state holds, into the "new" buffer.  No centrality update.  In an edge-partitioned run
 * each rank calls this on its own edge subset; the per-register max of all ranks' "new"
 * buffers (an all-reduce MAX) is the reference's "new" map. */
void hbo_dense_step_local(hbo_dense *s, int flags)
{
    const uint64_t n = s->n;
    const int frontier = (flags & HBO_FRONTIER) != 0;
    uint64_t active = 0, touched = 0;
#ifdef _OPENMP
    int nt = eff_threads(s->threads, (n ? s->row_ptr[n] : 0) + 8 * n);
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nt) reduction(+ : active, touched)
#endif
    for (int64_t vi = 0; vi < (int64_t)n; vi++) {
        const uint64_t v = (uint64_t)vi;
        const uint8_t *ov = s->old_regs + 64 * v;
        uint8_t acc[64];
        memcpy(acc, ov, 64); /* new[v] == old[v] on entry (Counters::step, harmonic.rs:210-212) */
        uint64_t act = 0;
        /* The merge is a per-register max, so edge order is irrelevant (App. A-5). */
        const uint64_t e_end = s->row_ptr[v + 1];
        for (uint64_t e = s->row_ptr[v]; e < e_end; e++) {
            uint32_t u = s->src[e];
            /* the gathers are random 64-byte reads: keep a few cache misses in flight */
            if (e + 8 < e_end) __builtin_prefetch(s->old_regs + 64 * (uint64_t)s->src[e + 8], 0, 0);
            if (s->changed_prev[u]) act++;
            else if (frontier) continue; /* bloom / exact set: results-inert (App. C-1) */
            const uint8_t *ou = s->old_regs + 64 * (uint64_t)u;
            for (int i = 0; i < 64; i++) acc[i] = ou[i] > acc[i] ? ou[i] : acc[i]; /* -O3: pmaxub */
        }
        memcpy(s->new_regs + 64 * v, acc, 64);
        active += act;
        touched += (act != 0);
    }
    s->last_active = active;
    s->last_touched = touched;
}

uint8_t *hbo_dense_pending_registers(hbo_dense *s) { return s->new_regs; }

/* Second half: changed detection, update_centralities (harmonic.rs:159-176), then
 * counters.step(); t += 1 (harmonic.rs:273-275). */
int hbo_dense_step_finish(hbo_dense *s, int flags, hbo_pass_stats *st)
{
    const uint64_t n = s->n;
    const int literal = (flags & HBO_LITERAL) != 0;
    const double denom = (double)(s->t + 1); /* (t + 1) as f64, harmonic.rs:174 */
    uint64_t changed = 0;
#ifdef _OPENMP
    int nt = eff_threads(s->threads, 8 * n);
#pragma omp parallel for schedule(static) num_threads(nt) reduction(+ : changed)
#endif
    for (int64_t vi = 0; vi < (int64_t)n; vi++) {
        const uint64_t v = (uint64_t)vi;
        const uint8_t *ov = s->old_regs + 64 * v;
        const uint8_t *nv = s->new_regs + 64 * v;
        int ch = memcmp(nv, ov, 64) != 0; /* "any from > to" happened at least once */
        s->changed_next[v] = (uint8_t)ch;
        changed += (uint64_t)ch;
        uint64_t sz_old = literal ? hbo_hll_size_ex(ov, s->bsearch, NULL, NULL) : s->size_old[v];
        uint64_t sz_new = (literal || ch) ? hbo_hll_size_ex(nv, s->bsearch, NULL, NULL) : sz_old;
        uint64_t d = (sz_new >= sz_old) ? sz_new - sz_old : 0; /* checked_sub().unwrap_or_default() */
        hbo_kahan_add(&s->ksum[v], &s->kerr[v], (double)d / denom);
        s->size_old[v] = sz_new;
    }
    uint8_t *tr = s->old_regs; s->old_regs = s->new_regs; s->new_regs = tr;
    /* every row of "new" is rewritten by the next step_local, so no clone is needed */
    uint8_t *tc = s->changed_prev; s->changed_prev = s->changed_next; s->changed_next = tc;
    s->has_changes = changed != 0;
    if (st) {
        st->pass = s->t;
        st->active_edges = s->last_active;
        st->touched = s->last_touched;
        st->changed = changed;
        st->has_changes = s->has_changes;
    }
    s->t += 1;
    return s->has_changes;
}

/* One pass of the loop body harmonic.rs:237-275. */
int hbo_dense_step(hbo_dense *s, int flags, hbo_pass_stats *st)
{
    hbo_dense_step_local(s, flags);
    return hbo_dense_step_finish(s, flags, st);
}

uint64_t hbo_dense_run(hbo_dense *s, int flags)
{
    /* harmonic.rs:237-240: loop { if !has_changes { break } ... } */
    while (s->has_changes) hbo_dense_step(s, flags, NULL);
    return s->t;
}

const uint8_t *hbo_dense_registers(const hbo_dense *s) { return s->old_regs; }
const double *hbo_dense_kahan_sum(const hbo_dense *s) { return s->ksum; }
const double *hbo_dense_kahan_err(const hbo_dense *s) { return s->kerr; }
const uint64_t *hbo_dense_sizes(const hbo_dense *s) { return s->size_old; }
uint64_t hbo_dense_passes(const hbo_dense *s) { return s->t; }

uint64_t hbo_dense_finish(const hbo_dense *s, double *out, uint8_t *keep)
{
    /* harmonic.rs:229 norm_factor = (num_nodes - 1) as f64; :178-195 */
    const double norm = (double)(s->n - 1);
    uint64_t k = 0;
    for (uint64_t v = 0; v < s->n; v++) {
        double c = s->ksum[v]; /* f64::from(KahanSum) = sum (kahan_sum.rs:35-39) */
        int kp = c > 0.0;
        double r = 0.0;
        if (kp) {
            r = c / norm;
            if (!isfinite(r)) r = 0.0;
            k++;
        }
        if (out) out[v] = r;
        if (keep) keep[v] = (uint8_t)kp;
    }
    return k;
}


/* Order-independent checksum of the state after the last executed pass (test infrastructure for
 * per-pass parity at sizes where register arrays are too big to ship around): out[0] over the
 * registers, out[1] over the Kahan (sum, err) bit patterns; node v contributes a 64-bit mix of
 * (v, its words), contributions are added mod 2^64.  The GPU library computes the same function
 * (hb_debug_state_hash). */
static inline uint64_t hash_mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

void hbo_dense_state_hash(const hbo_dense *s, uint64_t out[2])
{
    uint64_t hr = 0, hk = 0;
#pragma omp parallel for schedule(static) reduction(+ : hr, hk) num_threads(eff_threads(s->threads, 8 * s->n))
    for (int64_t v = 0; v < (int64_t)s->n; v++) {
        uint64_t r = (uint64_t)v * 0x9E3779B97F4A7C15ull + 1ull;
        for (int k = 0; k < 8; k++) {
            uint64_t w;
            memcpy(&w, s->old_regs + 64 * (uint64_t)v + 8 * k, 8);
            r = hash_mix64(r ^ w);
        }
        hr += r;
        uint64_t a, b;
        memcpy(&a, &s->ksum[v], 8);
        memcpy(&b, &s->kerr[v], 8);
        hk += hash_mix64(hash_mix64(((uint64_t)v + 0x632BE59BD9B4E019ull) ^ a) ^ b);
    }
    out[0] = hr;
    out[1] = hk;
}

/* ------------------------------------------------------------------------------- */
/* structure-faithful path                                                           */
/* ------------------------------------------------------------------------------- */

static inline u128 to_u128(hbo_u128 x) { return ((u128)x.hi << 64) | x.lo; }

static int cmp_u128(const void *a, const void *b)
{
    u128 x = *(const u128 *)a, y = *(const u128 *)b;
    return (x > y) - (x < y);
}

/* bloom/src/lib.rs:36-41 */
uint64_t hbo_bloom_num_bits(uint64_t estimated_items, double fp)
{
    double l2 = log(2.0);
    return (uint64_t)ceil(((double)estimated_items) * log(fp) / (-8.0 * (l2 * l2)));
}

/* bloom/src/lib.rs:108-123 */
uint64_t hbo_bloom_estimate_card(uint64_t num_bits, uint64_t num_ones)
{
    if (num_ones == 0 || num_bits == 0) return 0;
    if (num_ones == num_bits) return UINT64_MAX;
    /* (-(len as i64) * (1.0 - ones/len).ln() as i64).try_into().unwrap_or_default()
     * `as i64` binds tighter than `*`: the logarithm is truncated first. */
    double l = log(1.0 - (double)num_ones / (double)num_bits);
    int64_t li = (l != l) ? 0 : (l <= -9223372036854775808.0 ? INT64_MIN : (int64_t)l);
    int64_t r = -(int64_t)num_bits * li;
    return r < 0 ? 0 : (uint64_t)r;
}

typedef struct {
    uint64_t num_bits;
    uint64_t *words;
} bloom_t;

static bloom_t bloom_new(uint64_t n_items)
{
    bloom_t b;
    b.num_bits = hbo_bloom_num_bits(n_items, 0.05); /* harmonic.rs:221,242 */
    b.words = (uint64_t *)calloc((b.num_bits + 63) / 64 + 1, 8);
    return b;
}
static inline uint64_t bloom_slot(const bloom_t *b, u128 id)
{
    /* insert_u128/contains_u128 use the low 64 bits (bloom/src/lib.rs:91-102) */
    uint64_t h = (uint64_t)id * 11400714819323198549ull;
    return h % b->num_bits;
}
static inline void bloom_insert(bloom_t *b, u128 id)
{
    uint64_t s = bloom_slot(b, id);
    b->words[s >> 6] |= 1ull << (s & 63);
}
static inline int bloom_contains(const bloom_t *b, u128 id)
{
    uint64_t s = bloom_slot(b, id);
    return (int)((b->words[s >> 6] >> (s & 63)) & 1);
}
static uint64_t bloom_estimate_card(const bloom_t *b)
{
    uint64_t ones = 0;
    for (uint64_t i = 0; i < (b->num_bits + 63) / 64; i++) ones += (uint64_t)__builtin_popcountll(b->words[i]);
    return hbo_bloom_estimate_card(b->num_bits, ones);
}

/* open-addressing set of (from,to) pairs: itertools::unique_by((from,to)), store.rs:313 */
typedef struct { u128 from, to; } pair_t;
typedef struct {
    pair_t *slots;
    uint8_t *used;
    uint64_t mask;
} pairset_t;

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
static int pairset_init(pairset_t *ps, uint64_t cap_items)
{
    uint64_t c = 16;
    while (c < cap_items * 2) c <<= 1;
    ps->slots = (pair_t *)malloc(c * sizeof(pair_t));
    ps->used = (uint8_t *)calloc(c, 1);
    ps->mask = c - 1;
    return ps->slots && ps->used;
}
static void pairset_clear(pairset_t *ps) { memset(ps->used, 0, ps->mask + 1); }
/* returns 1 if newly inserted (first occurrence) */
static inline int pairset_insert(pairset_t *ps, u128 from, u128 to)
{
    uint64_t h = mix64((uint64_t)from ^ mix64((uint64_t)(from >> 64) + 0x9e3779b97f4a7c15ull)) ^
                 mix64((uint64_t)to * 3 + mix64((uint64_t)(to >> 64)));
    uint64_t i = h & ps->mask;
    while (ps->used[i]) {
        if (ps->slots[i].from == from && ps->slots[i].to == to) return 0;
        i = (i + 1) & ps->mask;
    }
    ps->used[i] = 1;
    ps->slots[i].from = from;
    ps->slots[i].to = to;
    return 1;
}

/* BTreeMap<NodeID, _>::get analogue: ordered lookup by id */
static inline int64_t node_find(const u128 *ids, uint64_t n, u128 key)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (ids[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && ids[lo] == key) ? (int64_t)lo : -1;
}

typedef struct { uint32_t from, to; } iedge_t;
static int cmp_iedge_from(const void *a, const void *b)
{
    const iedge_t *x = (const iedge_t *)a, *y = (const iedge_t *)b;
    if (x->from != y->from) return (x->from > y->from) - (x->from < y->from);
    return (x->to > y->to) - (x->to < y->to);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline int regs_any_greater(const uint8_t *from, const uint8_t *to)
{
    for (int i = 0; i < 64; i++) if (from[i] > to[i]) return 1; /* harmonic.rs:94-98,137-141 */
    return 0;
}

uint64_t hbo_faithful_run(const hbo_edge *edges, uint64_t m, hbo_u128 *out_ids, double *out_vals,
                          uint64_t cap, hbo_faithful_stats *stats)
{
    return hbo_faithful_run_pages(edges, m, NULL, 0, out_ids, out_vals, cap, stats);
}

uint64_t hbo_faithful_run_pages(const hbo_edge *edges, uint64_t m, const hbo_edge *pages, uint64_t mp,
                                hbo_u128 *out_ids, double *out_vals, uint64_t cap, hbo_faithful_stats *stats)
{
    return hbo_faithful_run_segments(edges, m, pages, mp, NULL, 0, out_ids, out_vals, cap, stats);
}

/* LinksScorer, crates/core/src/webgraph/query/raw/links.rs:115-232 - what a ForwardlinksQuery yields from ONE
 * segment's posting list of the term from_id = self.  to[i] = ToId (the de-duplication column, forwardlink.rs:99-101)
 * of the list's i-th document in doc order; emit[i] := 1 for the documents the scorer yields.
 *   new() (:143-165): leading self links are skipped; last_dedup_val = to of the first document left.
 *   advance() (:199-229): postings.advance(); WHILE has_seen(last doc of the current 128-document block) the whole
 *   block is jumped (block_cursor.advance + reset_cursor_start_block, :203-213) - the skip entry only exists for full
 *   blocks, the final partial block reports TERMINATED (tantivy postings/skip.rs:122-126,276-282), whose dedup value is
 *   None; then documents are skipped while has_seen(doc) || skip_self(doc); last_dedup_val = to of the document
 *   reached.  has_seen = "equals the LAST yielded to" only (:186-190): the de-duplication is adjacent, not global. */
void hbo_links_scorer(const hbo_u128 *to, uint64_t len, hbo_u128 self, uint8_t *emit)
{
    const uint64_t B = 128; /* COMPRESSION_BLOCK_SIZE */
    const uint64_t full = (len / B) * B;
    const u128 me = to_u128(self);
    for (uint64_t i = 0; i < len; i++) emit[i] = 0;
    uint64_t pos = 0;
    while (pos < len && to_u128(to[pos]) == me) pos++;
    if (pos >= len) return;
    u128 last = to_u128(to[pos]);
    while (pos < len) {
        emit[pos] = 1;
        pos++;
        while (pos < full && to_u128(to[(pos / B) * B + B - 1]) == last) pos = (pos / B) * B + B;
        while (pos < len && (to_u128(to[pos]) == last || to_u128(to[pos]) == me)) pos++;
        if (pos < len) last = to_u128(to[pos]);
    }
}

typedef struct { uint32_t from; uint64_t idx; } pagedoc_t;
static int cmp_pagedoc(const void *a, const void *b)
{
    const pagedoc_t *x = (const pagedoc_t *)a, *y = (const pagedoc_t *)b;
    if (x->from != y->from) return x->from < y->from ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* doc order inside a posting list */
}

/* pages != NULL: update_changed_counters follows PAGE-level records like the reference does:
 * ForwardlinksQuery::new(host id) matches documents whose page-level from_id equals the host id and yields their
 * page-level to_id (query/forwardlink.rs:95-101,153-173); harmonic.rs:91-92 then looks both ends up in the
 * host-keyed maps.  `pages` = those (from_id, to_id, rel_flags) records (any superset of them: records whose
 * ends are not host nodes fall out at the map lookup).  pages == NULL: host-level semantics (the forward index of
 * the de-duplicated host edges). */
uint64_t hbo_faithful_run_segments(const hbo_edge *edges, uint64_t m, const hbo_edge *pages, uint64_t mp,
                                   const uint64_t *seg_len, uint64_t nseg, hbo_u128 *out_ids, double *out_vals, uint64_t cap,
                                   hbo_faithful_stats *stats)
{
    hbo_faithful_stats st;
    memset(&st, 0, sizeof(st));
    /* ---- host_nodes(): every endpoint of every doc, no flag filtering (store.rs:338-357) */
    u128 *ids = (u128 *)malloc((2 * m + 1) * sizeof(u128));
    for (uint64_t i = 0; i < m; i++) {
        ids[2 * i] = to_u128(edges[i].from);
        ids[2 * i + 1] = to_u128(edges[i].to);
    }
    qsort(ids, 2 * m, sizeof(u128), cmp_u128);
    uint64_t n = 0;
    for (uint64_t i = 0; i < 2 * m; i++)
        if (n == 0 || ids[n - 1] != ids[i]) ids[n++] = ids[i];
    st.n = n;
    if (n == 0) { /* empty graph: defined as empty result (App. C-9) */
        free(ids);
        if (stats) *stats = st;
        return 0;
    }
    /* ---- initialize (harmonic.rs:53-73): one heap Vec<u8> per counter, like the maps */
    uint8_t **old_c = (uint8_t **)malloc(n * sizeof(uint8_t *));
    uint8_t **new_c = (uint8_t **)malloc(n * sizeof(uint8_t *));
    double *ksum = (double *)calloc(n, 8), *kerr = (double *)calloc(n, 8);
    for (uint64_t v = 0; v < n; v++) {
        old_c[v] = (uint8_t *)calloc(64, 1);
        hbo_hll_add(old_c[v], (uint64_t)ids[v]);
        new_c[v] = (uint8_t *)malloc(64);
        memcpy(new_c[v], old_c[v], 64);
    }
    /* harmonic.rs:221-225 */
    bloom_t changed = bloom_new(n);
    for (uint64_t v = 0; v < n; v++) bloom_insert(&changed, ids[v]);
    /* harmonic.rs:228-229 */
    uint64_t threshold = (uint64_t)round(fmax(sqrt((double)n), 0.0));
    double norm = (double)(n - 1);
    int exact_counting = 0, has_changes = 1;
    uint64_t t = 0;
    uint8_t *exact_set = (uint8_t *)calloc(n, 1); /* BTreeSet<NodeID> as membership flags */
    uint64_t exact_len = 0;
    uint8_t *exact_next = (uint8_t *)calloc(n, 1);

    pairset_t ps;
    pairset_init(&ps, m);
    /* forward index used by update_changed_counters' ForwardlinksQuery (host-level
     * semantics); the reference has it on disk, so it is built once,
     * outside the timed loop. */
    iedge_t *fwd = (iedge_t *)malloc((m + 1) * sizeof(iedge_t));
    uint64_t m_unique = 0, m_eff = 0;
    for (uint64_t i = 0; i < m; i++) {
        u128 f = to_u128(edges[i].from), tt = to_u128(edges[i].to);
        if (!pairset_insert(&ps, f, tt)) continue;
        m_unique++;
        if (edges[i].rel_flags & HBO_SKIPPED_REL_MASK) continue;
        fwd[m_eff].from = (uint32_t)node_find(ids, n, f);
        fwd[m_eff].to = (uint32_t)node_find(ids, n, tt);
        m_eff++;
    }
    st.m_unique = m_unique;
    st.m_eff = m_eff;
    uint64_t fwd_len = m_eff;
    if (pages) {
        /* the tail follows what ForwardlinksQuery::new(host id) returns from the page-level documents instead
         * (harmonic.rs:82-87): per segment (seg_len[]; NULL = all documents are one segment) and host, the documents the
         * LinksScorer yields; THEN the rel filter on those (:87) and the two map lookups (:91-92) */
        free(fwd);
        fwd = (iedge_t *)malloc((mp + 1) * sizeof(iedge_t));
        fwd_len = 0;
        const uint64_t one = mp;
        if (!seg_len) { seg_len = &one; nseg = 1; }
        pagedoc_t *docs = (pagedoc_t *)malloc((mp + 1) * sizeof(pagedoc_t));
        hbo_u128 *tos = (hbo_u128 *)malloc((mp + 1) * sizeof(hbo_u128));
        uint8_t *emit = (uint8_t *)malloc(mp + 1);
        uint64_t base = 0;
        for (uint64_t sgi = 0; sgi < nseg && base < mp; sgi++) {
            const uint64_t cnt = seg_len[sgi] < mp - base ? seg_len[sgi] : mp - base;
            uint64_t nd = 0;
            for (uint64_t i = base; i < base + cnt; i++) { /* posting lists of the terms that are host node ids */
                int64_t ui = node_find(ids, n, to_u128(pages[i].from));
                if (ui < 0) continue;
                docs[nd].from = (uint32_t)ui;
                docs[nd].idx = i;
                nd++;
            }
            qsort(docs, nd, sizeof(pagedoc_t), cmp_pagedoc);
            for (uint64_t a = 0; a < nd;) {
                uint64_t b = a;
                while (b < nd && docs[b].from == docs[a].from) b++;
                for (uint64_t k = a; k < b; k++) tos[k - a] = pages[docs[k].idx].to;
                hbo_u128 self;
                self.lo = (uint64_t)ids[docs[a].from];
                self.hi = (uint64_t)(ids[docs[a].from] >> 64);
                hbo_links_scorer(tos, b - a, self, emit);
                for (uint64_t k = a; k < b; k++) {
                    const hbo_edge *pg = &pages[docs[k].idx];
                    if (!emit[k - a]) continue;
                    if (pg->rel_flags & HBO_SKIPPED_REL_MASK) continue; /* :87 */
                    int64_t vi = node_find(ids, n, to_u128(pg->to));
                    if (vi < 0) continue; /* :91-92: both lookups must hit */
                    fwd[fwd_len].from = docs[a].from;
                    fwd[fwd_len].to = (uint32_t)vi;
                    fwd_len++;
                }
                a = b;
            }
            base += cnt;
        }
        free(docs); free(tos); free(emit);
    }
    qsort(fwd, fwd_len, sizeof(iedge_t), cmp_iedge_from);
    uint64_t *fwd_ptr = (uint64_t *)calloc(n + 2, 8);
    for (uint64_t i = 0; i < fwd_len; i++) fwd_ptr[fwd[i].from + 1]++;
    for (uint64_t v = 0; v < n; v++) fwd_ptr[v + 1] += fwd_ptr[v];

    double t0 = now_s();
    for (;;) { /* harmonic.rs:237 */
        if (!has_changes) break;
        bloom_t new_changed = bloom_new(n); /* :242 */
        if (exact_len != 0 && exact_len <= threshold) {
            /* update_changed_counters (harmonic.rs:75-114) */
            st.passes_exact++;
            has_changes = 0;
            uint64_t new_len = 0;
            memset(exact_next, 0, n);
            for (uint64_t u = 0; u < n; u++) {
                if (!exact_set[u]) continue;
                for (uint64_t e = fwd_ptr[u]; e < fwd_ptr[u + 1]; e++) {
                    uint32_t v = fwd[e].to;
                    if (regs_any_greater(old_c[u], new_c[v])) {
                        hbo_hll_merge(new_c[v], old_c[u]);
                        bloom_insert(&new_changed, ids[v]);
                        if (!exact_next[v]) { exact_next[v] = 1; new_len++; }
                        has_changes = 1;
                    }
                }
            }
            uint8_t *tmp = exact_set; exact_set = exact_next; exact_next = tmp;
            exact_len = new_len;
        } else {
            /* update_all_counters (harmonic.rs:116-157); host_edges() re-streams and
             * re-deduplicates every pass (store.rs:297-314) */
            int track = exact_counting;
            if (track) { memset(exact_set, 0, n); exact_len = 0; }
            has_changes = 0;
            pairset_clear(&ps);
            for (uint64_t i = 0; i < m; i++) {
                u128 f = to_u128(edges[i].from), tt = to_u128(edges[i].to);
                if (!pairset_insert(&ps, f, tt)) continue;                  /* unique_by */
                if (edges[i].rel_flags & HBO_SKIPPED_REL_MASK) continue;    /* :131 */
                if (!bloom_contains(&changed, f)) continue;                 /* :133 */
                int64_t vi = node_find(ids, n, tt), ui = node_find(ids, n, f); /* :135 */
                if (vi < 0 || ui < 0) continue;
                if (regs_any_greater(old_c[ui], new_c[vi])) {
                    hbo_hll_merge(new_c[vi], old_c[ui]);
                    bloom_insert(&new_changed, tt);
                    if (track && !exact_set[vi]) { exact_set[vi] = 1; exact_len++; }
                    has_changes = 1;
                }
            }
        }
        /* update_centralities (harmonic.rs:159-176): every node, both sizes afresh */
        for (uint64_t v = 0; v < n; v++) {
            uint64_t sn = hbo_hll_size(new_c[v]), so = hbo_hll_size(old_c[v]);
            uint64_t d = sn >= so ? sn - so : 0;
            hbo_kahan_add(&ksum[v], &kerr[v], (double)d / (double)(t + 1));
        }
        /* counters.step(): old = new.clone() (harmonic.rs:210-212) - n fresh allocations */
        for (uint64_t v = 0; v < n; v++) {
            free(old_c[v]);
            old_c[v] = (uint8_t *)malloc(64);
            memcpy(old_c[v], new_c[v], 64);
        }
        free(changed.words);
        changed = new_changed;
        t += 1;
        if (bloom_estimate_card(&changed) <= threshold) exact_counting = 1; /* :277-279 */
    }
    st.seconds_loop = now_s() - t0;
    st.passes = t;

    /* normalize_centralities (harmonic.rs:178-195), ascending id */
    uint64_t k = 0;
    int overflow = 0;
    for (uint64_t v = 0; v < n; v++) {
        double c = ksum[v];
        if (!(c > 0.0)) continue;
        double r = c / norm;
        if (!isfinite(r)) r = 0.0;
        if (out_ids || out_vals) {
            if (k >= cap) { overflow = 1; break; }
            if (out_ids) { out_ids[k].lo = (uint64_t)ids[v]; out_ids[k].hi = (uint64_t)(ids[v] >> 64); }
            if (out_vals) out_vals[k] = r;
        }
        k++;
    }
    for (uint64_t v = 0; v < n; v++) { free(old_c[v]); free(new_c[v]); }
    free(old_c); free(new_c); free(ksum); free(kerr); free(ids); free(changed.words);
    free(exact_set); free(exact_next); free(ps.slots); free(ps.used); free(fwd); free(fwd_ptr);
    if (stats) *stats = st;
    return overflow ? (uint64_t)-1 : k;
}

/* ------------------------------------------------------------------------------- */
/* harmonic_rank: the second half of store_harmonic (centrality/mod.rs:92-103)        */
/* ------------------------------------------------------------------------------- */
/* ranks[j] = position of result j when the results (given in ascending NodeID order) are sorted
 * by (Reverse(SortableFloat(centrality)), NodeID): SortableFloat::cmp = f64::total_cmp
 * (crates/core/src/lib.rs:259-263). */
typedef struct { uint64_t key; uint64_t idx; } hbo_rank_item;

static uint64_t total_cmp_key(double v)
{
    /* f64::total_cmp: compare the bit patterns as sign-magnitude integers */
    uint64_t b;
    memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

static int rank_item_cmp(const void *pa, const void *pb)
{
    const hbo_rank_item *a = (const hbo_rank_item *)pa, *b = (const hbo_rank_item *)pb;
    if (a->key != b->key) return a->key > b->key ? -1 : 1; /* Reverse: larger centrality first */
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0); /* then NodeID ascending */
}

int hbo_rank_results(const double *vals, uint64_t k, uint64_t *ranks)
{
    hbo_rank_item *items = (hbo_rank_item *)malloc((k ? k : 1) * sizeof(*items));
    if (!items) return -1;
    for (uint64_t i = 0; i < k; i++) {
        items[i].key = total_cmp_key(vals[i]);
        items[i].idx = i;
    }
    qsort(items, k, sizeof(*items), rank_item_cmp);
    for (uint64_t r = 0; r < k; r++) ranks[items[r].idx] = r;
    free(items);
    return 0;
}
