// In-register radix-2 DIF transforms shared by the NTT passes (ntt.hip) and the FRI fold (stark.hip).
#pragma once
#include "gl.hip.h"
#include <type_traits>

namespace zk {

__host__ __device__ constexpr int bitrev_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// compile-time loop
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// exponent e of the twiddle w_{2*half}^j = 2^e (forward) or 2^-e (inverse), 2*half <= 64:
// w_64 = MG.0[6] = 2^39, w_{64/m} = 2^(39 m); e is taken mod 192 (the order of 2)
__host__ __device__ constexpr int tw_pow2_exp(int half, int j, bool inv) {
    int e = (39 * (32 / half) * j) % 192;
    return inv ? (192 - e) % 192 : e;
}

// 2^LOG-point DIF NTT in registers, natural order in; X[k] ends up in x[bitrev(k)].  LOG <= 6: every
// twiddle is +-2^e, applied as a shift (2^96 = -1 folds into the order of the subtraction).
template <int LOG, bool INV>
__device__ __forceinline__ void ntt_reg(u64 (&x)[1 << LOG]) {
    static_assert(LOG <= 6, "power-of-two twiddles exist up to order 64");
    constexpr int n = 1 << LOG;
    static_for<0, LOG>([&](auto LI) {
        constexpr int half = 1 << (LOG - 1 - decltype(LI)::value);
        static_for<0, half>([&](auto JI) {
            constexpr int j = decltype(JI)::value;
            constexpr int e = tw_pow2_exp(half, j, INV);
#pragma unroll
            for (int blk = 0; blk < n; blk += 2 * half) {
                const u64 a = x[blk + j], b = x[blk + j + half];
                x[blk + j] = gl::add(a, b);
                if constexpr (e >= 96) x[blk + j + half] = gl::mul_pow2<e - 96>(gl::sub(b, a));
                else                   x[blk + j + half] = gl::mul_pow2<e>(gl::sub(a, b));
            }
        });
    });
}

}  // namespace zk
