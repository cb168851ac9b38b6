// hb_estimator.hip.h - HyperLogLog<64>::size() (hyperloglog.rs:4484-4516, bias tables, linear counting) and KahanSum += (kahan_sum.rs:47-54) on the device, bit-exact.
// Part of the device code of stract_amd/csrc/hb_kernels.hip.h (included from there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hb_regs.hip.h"

namespace hbk {

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
// lc != NULL [r6]: the f64 sum is only built when some lane of the WAVE holds a counter whose size() is not decided by its zero count
// (hll_size_from tests the linear-counting exit first and never looks at the sum then): after pass 0, and for the cold majority of the
// rows in every pass, whole waves skip the 16 v_add_f64 per lane.  sum_out is 0.0 then.
__device__ __forceinline__ void hll_sum_quad(const uint4 &v, double &sum_out, uint32_t &zeros_out, uint32_t &big_out, const uint8_t *lc = nullptr)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t zeros = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // zero bytes of the word: bit 7 of every byte of z marks a zero byte
        const uint32_t z = ~(((w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[k] | 0x7F7F7F7Fu);
        zeros += __popc(z);
        // largest register of the word, as max over the 16-bit lanes' high bytes and low bytes
        mx = pkmax(mx, pkmax(w[k] & 0x00FF00FFu, (w[k] >> 8) & 0x00FF00FFu));
    }
    uint32_t big = ((mx & 0xFFFFu) > 47u || (mx >> 16) > 47u) ? 1u : 0u;
    // quad reduction (xor 1, xor 2)
    zeros += quad_perm<0xB1>(zeros);
    big |= quad_perm<0xB1>(big);
    zeros += quad_perm<0x4E>(zeros);
    big |= quad_perm<0x4E>(big);
    zeros_out = zeros;
    big_out = big;
    sum_out = 0.0;
    if (lc) {
        const bool by_zero_count = zeros != 0 && lc[zeros] != 0xFFu; // hll_size_from's first test
        if (!__ballot(!by_zero_count)) return;                        // wave-uniform
    }
    // The 16 terms 2^-r of this lane are added as doubles built from their exponent field (hi word = (1023 - r) << 20):
    // with every register <= 47 all partial sums are multiples of 2^-47 below 2^7, so these additions are exact in any
    // order - the same value as the reference's left fold; four v_add_f64 per word instead of 64-bit integer shifts / adds.
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t r = (w[k] >> (8 * b)) & 0xFFu;
            acc += __hiloint2double((int)((1023u << 20) - (r << 20)), 0); // r <= 255: the exponent field stays positive
        }
    }
    // quad reduction; the f64 sums stay exact for the same reason
    {
        uint32_t lo = (uint32_t)__double2loint(acc), hi = (uint32_t)__double2hiint(acc);
        acc += __hiloint2double((int)quad_perm<0xB1>(hi), (int)quad_perm<0xB1>(lo));
        lo = (uint32_t)__double2loint(acc); hi = (uint32_t)__double2hiint(acc);
        acc += __hiloint2double((int)quad_perm<0x4E>(hi), (int)quad_perm<0x4E>(lo));
    }
    sum_out = acc; // when big != 0 the value is unused - the fold is replayed in register order
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
    // :4504-4515 : linear counting wins iff v != 0 and 64 ln(64/v) <= 40 - a function of the zero count ALONE (v >= 35), so it is
    // tested FIRST: the division, the 159-entry search and the 6-NN walk below are dead work for such a counter, and after pass 0
    // (a node's counter = one register per in-neighbour + its own) that is most rows.  Same value either way: the reference
    // computes e_star and then discards it (hyperloglog.rs:4511-4515).  [r6, VERDICT r5 #2b]
    const uint32_t l = lc[zeros]; // zeros in 0..64
    if (zeros != 0 && l != 0xFFu) return (uint64_t)l;
    const double z = 1.0 / sum;                 // :4494
    const double e = (0.709 * 4096.0) * z;      // :4496  am() * m.powi(2) * z
    double e_star = e;
    if (e <= 320.0) e_star = e - estimate_bias(raw, bias, e); // :4498-4502
    return f64_as_usize(e_star);
}

// All 4 lanes of the quad call this with their uint4; all get the same result.
__device__ __forceinline__ uint64_t hll_size_quad(const uint4 &v, const double *raw, const double *bias,
                                                  const uint8_t *lc)
{
    double sum;
    uint32_t zeros, big;
    hll_sum_quad(v, sum, zeros, big, lc);
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

} // namespace hbk
