// hb_regs.hip.h - register-block helpers shared by the device code: one quad (4 lanes) holds one 64-byte
// HyperLogLog<64> counter, lane q the registers [16q, 16q+16) as a uint4; byte-wise max = HyperLogLog::merge
// (crates/core/src/hyperloglog.rs:4531-4535) on even/odd bytes with v_pk_max_u16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hbk {

// ---- quad helpers ---------------------------------------------------------------------
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
// bit g of the result = any lane of quad g voted (ballot folded 4:1); scalar ALU work
__device__ __forceinline__ uint32_t pack16(uint64_t b)
{
    b |= b >> 1;
    b |= b >> 2;
    b &= 0x1111111111111111ull;
    b = (b | (b >> 3)) & 0x0303030303030303ull;
    b = (b | (b >> 6)) & 0x000F000F000F000Full;
    b = (b | (b >> 12)) & 0x000000FF000000FFull;
    b = (b | (b >> 24)) & 0xFFFFull;
    return (uint32_t)b;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b)
{
    us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
    us2 z = __builtin_elementwise_max(x, y);
    return __builtin_bit_cast(uint32_t, z);
}

// 16 registers of one lane kept as even/odd bytes so that one merge is 1 AND + 2 v_pk_max_u16 per word: the even bytes
// are compared as clean 16-bit lanes (0x00FF00FF masked); the odd bytes as the HIGH bytes of the unmasked 16-bit lanes -
// a lane's maximum has the larger high byte whatever the low bytes are, so o[] carries garbage in its low bytes until
// acc_value() masks it once.
struct Acc {
    uint32_t e[4], o[4];
};
__device__ __forceinline__ void acc_zero(Acc &a)
{
#pragma unroll
    for (int k = 0; k < 4; k++) a.e[k] = a.o[k] = 0;
}
__device__ __forceinline__ void acc_merge(Acc &a, const uint4 &r)
{
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        a.e[k] = pkmax(a.e[k], w[k] & 0x00FF00FFu);
        a.o[k] = pkmax(a.o[k], w[k]);
    }
}
__device__ __forceinline__ uint4 acc_value(const Acc &a)
{
    return make_uint4(a.e[0] | (a.o[0] & 0xFF00FF00u), a.e[1] | (a.o[1] & 0xFF00FF00u), a.e[2] | (a.o[2] & 0xFF00FF00u),
                      a.e[3] | (a.o[3] & 0xFF00FF00u));
}
__device__ __forceinline__ bool u4_ne(const uint4 &a, const uint4 &b)
{
    return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) != 0;
}

} // namespace hbk
