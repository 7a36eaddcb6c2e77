// Carry-free accumulation of 64 x 64-bit products (shared by the Poseidon permutation and the evaluations at xi).
#pragma once
#include "gl.hip.h"

namespace zk {
using gl::add_word;

// Batched dot products: sum_j c_j * x_j mod p with the constants c_j split in three limbs of 22/22/20 bits (two LDS words
// each) and the state words x_j given as 32-bit halves.  Six 64-bit accumulators take the 22x32-bit partial products straight
// from v_mad_u64_u32 (n * 2^54 < 2^64 for n <= 512: no carries); ONE recombination and ONE reduction per dot product instead
// of a multiplication with a reduction per term.
struct Acc6 { u64 a00, a10, a20, a01, a11, a21; };   // a[i][h]: limb i of the constants x half h of the words
__device__ __forceinline__ void acc_zero(Acc6& A) { A.a00 = A.a10 = A.a20 = A.a01 = A.a11 = A.a21 = 0; }
__device__ __forceinline__ void acc_word(Acc6& A, u64 s) { A.a00 = (u32)s; A.a01 = s >> 32; A.a10 = A.a20 = A.a11 = A.a21 = 0; }   // 1 * s
__device__ __forceinline__ void acc_mac(Acc6& A, const ulonglong2 v /* a split constant */, u32 x0, u32 x1) {
    const u32 l0 = (u32)v.x, l1 = (u32)(v.x >> 32), l2 = (u32)v.y;
    A.a00 += (u64)l0 * x0; A.a10 += (u64)l1 * x0; A.a20 += (u64)l2 * x0;
    A.a01 += (u64)l0 * x1; A.a11 += (u64)l1 * x1; A.a21 += (u64)l2 * x1;
}
__device__ __forceinline__ void acc_mac(Acc6& A, const u64* __restrict__ c /* LDS, 2 words */, u32 x0, u32 x1) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(c);
    const u32 l0 = (u32)v.x, l1 = (u32)(v.x >> 32), l2 = (u32)v.y;
    A.a00 += (u64)l0 * x0; A.a10 += (u64)l1 * x0; A.a20 += (u64)l2 * x0;
    A.a01 += (u64)l0 * x1; A.a11 += (u64)l1 * x1; A.a21 += (u64)l2 * x1;
}
// a * b + c as ONE v_mad_u64_u32, whatever a, b, c are: written as C, a product by a power of two becomes a 64-bit shift, an
// and-mask and an addition, and "c + (x >> 32)" a zero-extending move and a 64-bit addition.
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "s"(b), "v"(c));
    return d;
}
__device__ __forceinline__ u64 mul32(u32 a, u32 b) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b));
    return d;
}
// V = sum_{i,h} a[i][h] 2^(22 i + 32 h).  The six accumulators sit at bit offsets 0, 22, 44 = 32 + 12, 32, 54 = 32 + 22 and
// 76 = 64 + 12: their 32-bit halves, shifted by 22 or 12 bits, are accumulated by multiply-adds into four 64-bit columns
// Z0..Z3 spaced 32 bits apart (every column < 2^61), three more multiply-adds carry each column's high word into the next,
// and with t = 2^32, t^2 = t - 1, t^3 = -1:  V = (w0 + w1 t) + w2 (2^32 - 1) - Y3  -- 11 multiply-adds and one reduction
// where the 128-bit shifts and additions of the first version took 56 instructions.  Any u64 in, nc out.
__device__ __forceinline__ u64 acc_finish(const Acc6& A) {
    constexpr u32 S22 = 1u << 22, S12 = 1u << 12;
    const u64 Z0 = mad32((u32)A.a10, S22, A.a00);
    u64 Z1 = mad32((u32)(A.a10 >> 32), S22, A.a01);
    Z1 = mad32((u32)A.a11, S22, Z1);
    Z1 = mad32((u32)A.a20, S12, Z1);
    u64 Z2 = mul32((u32)(A.a11 >> 32), S22);
    Z2 = mad32((u32)(A.a20 >> 32), S12, Z2);
    Z2 = mad32((u32)A.a21, S12, Z2);
    const u64 Z3 = mul32((u32)(A.a21 >> 32), S12);
    const u64 Y1 = add_word(Z1, (u32)(Z0 >> 32));
    const u64 Y2 = add_word(Z2, (u32)(Y1 >> 32));
    const u64 Y3 = add_word(Z3, (u32)(Y2 >> 32));                     // < 2^42: the words above 2^96 are subtracted as one number
    u32 b0, b1, e0, e1;
    u32 t0 = __builtin_subc((u32)Z0, (u32)Y3, 0u, &b0);               // t = (w0 + w1 t) - Y3
    u32 t1 = __builtin_subc((u32)Y1, (u32)(Y3 >> 32), b0, &b1);
    const u32 mb = 0u - b1;                                           // borrowed: t -= 2^32 - 1 (t >= 2^64 - 2^42: cannot borrow again)
    t0 = __builtin_subc(t0, mb, 0u, &e0);
    t1 = __builtin_subc(t1, 0u, e0, &e1);
    return gl::mad_eps_nc((u32)Y2, gl::mk64(t0, t1));
}

}  // namespace zk
