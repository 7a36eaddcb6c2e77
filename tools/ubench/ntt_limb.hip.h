// In-register radix-2^LOG transforms on 24-bit limbs: the butterflies of an NTT pass without carry chains.
//
// On gfx950 an add-with-carry costs what a 64-bit multiply-add costs (4.2 cycles per wave instruction; a plain 32-bit
// add or subtract 2.4, profiles/r02/ubench_valu.txt), and a carry consumed as data adds wait states: a 64-bit modular
// butterfly is ~11 such instructions plus ~10 for its shift twiddle (ntt_reg.hip.h).  Here a word x < 2^64 is split into
// four signed limbs (l0, l1, l2, l3) in 32-bit registers, value = sum l_i 2^(24 i), taken modulo 2^96 + 1 = p (2^32 + 1):
//   * a + b, a - b          = four plain 32-bit adds / subtracts, no carries: 24-bit limbs leave 7 bits of headroom,
//                             enough for the four levels of a radix-16 transform;
//   * times 2^(24 q), -1    = a limb rotation with 2^96 = -1 at the wrap: folded into WHICH subtraction is emitted
//                             (b_i - a_i instead of a_i - b_i), no instruction at all;
//   * times 2^12            = four and / shift / shift-add triples (only the four odd twiddles of a radix-16's first level
//                             need it: every twiddle of a transform of <= 16 points is 2^(12 m));
//   * back to a word        = a bias K = 0 (mod p) with limbs >= 2^28, added to element 0 alone, reaches every output
//                             with coefficient one (element 0 only ever meets the twiddle 2^0), so all output limbs are
//                             non-negative; three multiply-adds then carry them into four 32-bit words and the usual
//                             2^64 = 2^32 - 1, 2^96 = -1 fold (gl::reduce_words) finishes.
// Same values as ntt_reg<LOG, INV>: tools/ubench/ubench_ntt16.hip runs both on the same words and counts mismatches (none).
#pragma once
#include "ntt_reg.hip.h"

namespace zk {

// K = sum k_i 2^(24 i) = 0 (mod p), 2^28 <= k_i < 2^29  (tools/ntt_limb_bias.py)
__device__ constexpr int NTT_LIMB_BIAS[4] = {0x107db9e4, 0x1f463b4e, 0x1cc4a163, 0x1f5e7d9d};

// (a, b) <- (a + b, (a - b) * 2^E), E in [0, 192), 12 | E
template <int E>
__device__ __forceinline__ void limb_bfly(int (&a)[4], int (&b)[4]) {
    constexpr bool neg = E >= 96;
    constexpr int e = E % 96, q = e / 24, r = e % 24;
    static_assert(r == 0 || r == 12, "twiddles of a transform of <= 16 points are powers of 2^12");
    int d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ip = (i + q) % 4;
        const bool flip = ((i + q) >= 4) != neg;      // wrapped past 2^96 = -1, or E >= 96
        d[ip] = flip ? b[i] - a[i] : a[i] - b[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] += b[i];
    if constexpr (r == 12) {                           // d * 2^12: d_i = lo_i + hi_i 2^12 -> (lo_i << 12) + hi_(i-1), -hi_3 at the wrap
        int lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = d[i] & 0xFFF; hi[i] = d[i] >> 12; }
        b[0] = (lo[0] << 12) - hi[3];
#pragma unroll
        for (int i = 1; i < 4; ++i) b[i] = (lo[i] << 12) + hi[i - 1];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = d[i];
    }
}

// 2^LOG-point DIF NTT, natural order in, X[k] in x[bitrev(k)] -- the contract of ntt_reg.  In: any u64; out: canonical
// words if CANON, else some u64 congruent to the value (what gl::mul accepts).
template <int LOG, bool INV, bool CANON>
__device__ __forceinline__ void ntt_reg_limb(u64 (&x)[1 << LOG]) {
    static_assert(LOG >= 1 && LOG <= 4, "headroom: 24 + LOG + 1 bits per limb");
    constexpr int n = 1 << LOG;
    int L[n][4];
#pragma unroll
    for (int i = 0; i < n; ++i) {
        L[i][0] = (int)((u32)x[i] & 0xFFFFFFu);
        L[i][1] = (int)((u32)(x[i] >> 24) & 0xFFFFFFu);
        L[i][2] = (int)(u32)(x[i] >> 48);
        L[i][3] = 0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) L[0][i] += NTT_LIMB_BIAS[i];
    static_for<0, LOG>([&](auto LI) {
        constexpr int half = 1 << (LOG - 1 - decltype(LI)::value);
        static_for<0, half>([&](auto JI) {
            constexpr int j = decltype(JI)::value;
            constexpr int e = tw_pow2_exp(half, j, INV);
#pragma unroll
            for (int blk = 0; blk < n; blk += 2 * half) limb_bfly<e>(L[blk + j], L[blk + j + half]);
        });
    });
#pragma unroll
    for (int i = 0; i < n; ++i) {
        // limbs in [0, 2^30): carry them into 32-bit words.  T0 < 2^55, T1 < 2^47, T2 < 2^39
        const u64 T0 = (u64)(u32)L[i][1] * (1u << 24) + (u32)L[i][0];
        const u64 T1 = (u64)(u32)L[i][2] * (1u << 16) + (u32)(T0 >> 32);
        const u64 T2 = (u64)(u32)L[i][3] * (1u << 8) + (u32)(T1 >> 32);
        x[i] = CANON ? gl::reduce_words((u32)T0, (u32)T1, (u32)T2, (u32)(T2 >> 32))
                     : gl::reduce_words_nc((u32)T0, (u32)T1, (u32)T2, (u32)(T2 >> 32));
    }
}

}  // namespace zk
