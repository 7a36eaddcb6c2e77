// In-register radix-2 DIF transforms shared by the NTT passes (ntt.hip) and the FRI fold (stark.hip).
#pragma once
#include "gl.cuh"

namespace zk {

__host__ __device__ constexpr int bitrev_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// 2^LOG-point DIF NTT in registers, natural order in; X[k] ends up in x[bitrev(k)].
// Twiddle w_{2h}^j = w_256^(j * 128/h): uniform addresses -> scalar loads.
template <int LOG>
__device__ __forceinline__ void ntt_reg(u64 (&x)[1 << LOG], const u64* __restrict__ w256) {
    constexpr int n = 1 << LOG;
#pragma unroll
    for (int lh = LOG - 1; lh >= 0; --lh) {
        const int half = 1 << lh;
#pragma unroll
        for (int blk = 0; blk < n; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                u64 a = x[blk + j], b = x[blk + j + half];
                x[blk + j] = gl::add(a, b);
                u64 d = gl::sub(a, b);
                x[blk + j + half] = (j == 0) ? d : gl::mul(d, w256[j * (128 / half)]);
            }
        }
    }
}

}  // namespace zk
