// Products of a constant matrix of full-width Goldilocks constants with a vector of field elements PER LANE, on the matrix pipe (round 6).
//
// out[o] = sum_j c[o][j] x[j] mod p for up to 4 TL x 4 KS constants c, where every lane of a wave holds its own x[] in registers and
// all lanes share c: the dense linear layers of Poseidon-Goldilocks (poseidon.hip: the pre-sparse matrix P of poseidon_opt.rs:121-131 and
// the two dense products of each lazy block of partial rounds, 12 x 12: KS = TL = 3).  The shapes are template parameters because the
// same form was measured for the radix-16 butterfly networks of an NTT pass (a 16-point DFT over a lane's sixteen points IS such a
// product, 16 x 16: KS = TL = 4) -- bit-exact and NOT faster there: tools/experiments/ntt_mfma.patch, profiles/r06/ntt_mfma.md.
// On the vector pipe a term costs six v_mad_u64_u32 and an output a 25-instruction recombination; here the
// 64 lanes' vectors are the B operand of v_mfma_i32_32x32x32_i8 and the constants the A operand:
//   * a word is 8 bytes x_b; c x = sum_b x_b (c 2^(8 b) mod p), and each (c 2^(8 b) mod p) is written in eight balanced base-256
//     digits a_d in [-128, 127] (its representative in (-p/2, p/2) always fits).  Row (o, d) of A holds digit d of every (j, b): the
//     product's row is the d-th byte column of the output, eight i32 columns per output (|column| <= 2^21), NOT the fifteen of a plain
//     byte-limb product -- the reduction mod p happened in the table;
//   * B wants signed bytes: x_b xor 0x80 = x_b - 128; the missing 128 sum(a) is a constant per output and rides, with a bias that makes
//     everything positive, on the two 64-bit addends of the recombination (K below);
//   * one 32 x 32 x 32 tile = 4 outputs x 8 digits against 4 words x 8 bytes for 32 lanes' vectors.  The rows are ordered so that the
//     sixteen accumulators of a lane are the 2 x 8 digit columns of two outputs of one vector; the wave's 64 vectors are
//     two column tiles, lanes l and l + 32 trade words by v_permlane32_swap on the way in (each supplies half of the K range) and
//     on the way out (each recombines two of a tile's four outputs for both of them).
// Per 12 x 12 product: 18 MFMAs (32 cycles each on the SIMD's matrix pipe, beside the other waves' vector work), 9 LDS fragment
// reads, 24 xors, 24 swaps and 12 recombinations of ~14 vector instructions; per 16 x 16 product 32 MFMAs and 16 recombinations.
#pragma once
#include "gl.hip.h"
#include "acc6.hip.h"
#include "ntt_reg.hip.h"   // static_for
#include <vector>
#include <string>
#include <cstring>
#include <type_traits>

namespace zk {
namespace pmfma {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// table of one product with KS k-steps (4 input words each) and TL tiles (4 outputs each), u64 words: the A fragments
// [tile][kstep][lane] x 16 bytes, then (KL, KH) per output
__host__ __device__ constexpr int frag_words(int KS, int TL) { return TL * KS * 64 * 2; }
__host__ __device__ constexpr int tab_words(int KS, int TL) { return frag_words(KS, TL) + 4 * TL * 2; }
constexpr int FRAG_WORDS = frag_words(3, 3);   // the 12 x 12 products of Poseidon: 9 KB
constexpr int TAB_WORDS = tab_words(3, 3);

// ---- host: the tables of one product -------------------------------------------------------------------------------------------
// coef[o * n_in + j] canonical, n_out <= 4 TL, n_in <= 4 KS; addend[o] (may be null): a canonical constant added to output o for free.
// Returns false if a column could exceed the bound the recombination assumes.
inline bool build_tables(const u64* coef, int n_out, int n_in, const u64* addend, u64* tab /* tab_words(KS, TL) */, int KS = 3, int TL = 3) {
    typedef unsigned __int128 u128;
    typedef __int128 i128;
    if (n_out > 4 * TL || n_in > 4 * KS || KS > 4 || TL > 4) return false;
    std::memset(tab, 0, sizeof(u64) * tab_words(KS, TL));
    signed char* fr = reinterpret_cast<signed char*>(tab);
    std::vector<signed char> digv(16 * 16 * 64, 0);
    auto dig = [&](int o, int j, int b, int d) -> signed char& { return digv[((o * 16 + j) * 8 + b) * 8 + d]; };   // [o][j][b][d]
    const i128 P = (i128)GL_P;
    const i128 S = (i128)(~0ull) / 255;        // (256^8 - 1) / 255
    for (int o = 0; o < n_out; ++o)
        for (int j = 0; j < n_in; ++j) {
            u64 v = coef[o * n_in + j] % GL_P;
            for (int b = 0; b < 8; ++b) {
                i128 s = (i128)v;
                if (s > 127 * S) s -= P;
                if (s < -128 * S) return false;
                for (int d = 0; d < 8; ++d) {
                    int lowb = (int)(((s % 256) + 256) % 256);
                    int dg = lowb >= 128 ? lowb - 256 : lowb;
                    dig(o, j, b, d) = (signed char)dg;
                    s = (s - dg) / 256;
                }
                if (s != 0) return false;
                v = (u64)(((u128)v << 8) % GL_P);   // c 2^(8 (b + 1)) mod p
            }
        }
    // fragments: tile t, k-step s, lane l, byte i  <-  row m = l & 31, K slot (l >> 5, i)
    for (int t = 0; t < TL; ++t)
        for (int s = 0; s < KS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int m = l & 31, H = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);
                const int o = 4 * t + 2 * H + (r >> 3), d = r & 7, h = l >> 5;
                for (int i = 0; i < 16; ++i) {
                    const int j = 4 * s + 2 * h + (i >> 3), b = i & 7;
                    fr[((t * KS + s) * 64 + l) * 16 + i] = (o < n_out && j < n_in) ? dig(o, j, b, d) : 0;
                }
            }
    // K: the bytes reach the matrix pipe as x_b - 128; what is missing is 128 * (sum of the row's digits) per column, a constant.
    for (int o = 0; o < 4 * TL; ++o) {
        i128 corr = 0;
        long long worst = 0;
        if (o < n_out)
            for (int d = 0; d < 8; ++d) {
                long long sa = 0, sabs = 0;
                for (int j = 0; j < n_in; ++j)
                    for (int b = 0; b < 8; ++b) { sa += dig(o, j, b, d); sabs += dig(o, j, b, d) < 0 ? -dig(o, j, b, d) : dig(o, j, b, d); }
                corr += ((i128)(128 * sa)) << (8 * d);
                if (128 * sabs > worst) worst = 128 * sabs;
            }
        if (worst > (1ll << 21)) return false;
        i128 c = corr % P; if (c < 0) c += P;
        if (addend && o < n_out) c = (c + (i128)(addend[o] % GL_P)) % P;
        // KL + 2^32 KH = c (mod p), both in [2^47, 2^47 + 2^32)
        i128 tau = (c - ((i128)1 << 47) - (((i128)1 << 79) % P)) % P; if (tau < 0) tau += P;
        const u64 tv = (u64)tau;
        tab[frag_words(KS, TL) + 2 * o] = (1ull << 47) + (tv & 0xFFFFFFFFull);
        tab[frag_words(KS, TL) + 2 * o + 1] = (1ull << 47) + (tv >> 32);
    }
    return true;
}

// Host-side check of a table against the coefficients it was built from (no GPU): every row's digits spell (c 2^(8 b) mod p) or its
// negative-side representative, the fragments sit where product_t() reads them, and the two addends of every output are in range and
// congruent to 128 * (sum of the row's digits, weighted) + addend + the bias.  Returns "" or what is wrong.
inline std::string check_tables(const u64* coef, int n_out, int n_in, const u64* addend, const u64* tab, int KS = 3, int TL = 3) {
    typedef __int128 i128;
    const signed char* fr = reinterpret_cast<const signed char*>(tab);
    const i128 P = (i128)GL_P;
    auto frag = [&](int o, int d, int j, int b) -> int {   // the byte product_t() multiplies word j's byte b with for output o's digit column d
        const int t = o / 4, H = (o % 4) / 2, r = 8 * (o % 2) + d, m = (r & 3) + 8 * (r >> 2) + 4 * H;
        const int s = j / 4, h = (j % 4) / 2, i = 8 * (j % 2) + b;
        return fr[((t * KS + s) * 64 + (32 * h + m)) * 16 + i];
    };
    for (int o = 0; o < 4 * TL; ++o) {
        i128 corr = 0;
        for (int j = 0; j < 4 * KS; ++j)
            for (int b = 0; b < 8; ++b) {
                i128 v = 0;
                for (int d = 7; d >= 0; --d) v = v * 256 + frag(o, d, j, b);
                corr += 128 * v;
                i128 want = 0;
                if (o < n_out && j < n_in) { want = (i128)(coef[o * n_in + j] % GL_P); for (int k = 0; k < b; ++k) want = (want << 8) % P; }
                i128 got = v % P; if (got < 0) got += P;
                if (got != want) return "digits of output " + std::to_string(o) + ", word " + std::to_string(j) + ", byte " + std::to_string(b) + " do not spell c 2^(8 b) mod p";
            }
        const u64 KL = tab[frag_words(KS, TL) + 2 * o], KH = tab[frag_words(KS, TL) + 2 * o + 1];
        if (KL < (1ull << 47) || KL >= (1ull << 47) + (1ull << 32) || KH < (1ull << 47) || KH >= (1ull << 47) + (1ull << 32)) return "addends of output " + std::to_string(o) + " out of range";
        i128 want = corr % P; if (want < 0) want += P;
        if (addend && o < n_out) want = (want + (i128)(addend[o] % GL_P)) % P;
        const i128 got = ((i128)KL + (((i128)KH << 32) % P)) % P;
        if (got != want) return "addends of output " + std::to_string(o) + " are not congruent to the bias correction";
    }
    return "";
}

// What the device computes from a table, step by step on the host (the i32 columns, the two biased 64-bit sums, the fold): out[o] mod p
// for a vector x of ANY u64 words.  With check_tables() and the known-answer tests on the device this pins the form on both sides.
inline void emulate_product(const u64* tab, const u64* x, int n_words, u64* out, int KS = 3, int TL = 3) {
    typedef unsigned __int128 u128;
    const signed char* fr = reinterpret_cast<const signed char*>(tab);
    for (int o = 0; o < 4 * TL; ++o) {
        long long col[8];
        for (int d = 0; d < 8; ++d) {
            long long c = 0;
            for (int j = 0; j < 4 * KS; ++j)
                for (int b = 0; b < 8; ++b) {
                    const int t = o / 4, H = (o % 4) / 2, r = 8 * (o % 2) + d, m = (r & 3) + 8 * (r >> 2) + 4 * H;
                    const int s = j / 4, h = (j % 4) / 2, i = 8 * (j % 2) + b;
                    const int a = fr[((t * KS + s) * 64 + (32 * h + m)) * 16 + i];
                    const int xb = (int)(signed char)((unsigned char)((j < n_words ? x[j] : 0) >> (8 * b)) ^ 0x80);
                    c += (long long)a * xb;
                }
            col[d] = c;                                        // |c| <= 2^21: fits the pipe's i32
        }
        const u64 KL = tab[frag_words(KS, TL) + 2 * o], KH = tab[frag_words(KS, TL) + 2 * o + 1];
        const u64 lo = (u64)((long long)KL + col[0] + (col[1] << 8) + (col[2] << 16) + (col[3] << 24));
        const u64 hi = (u64)((long long)KH + col[4] + (col[5] << 8) + (col[6] << 16) + (col[7] << 24));
        out[o] = (u64)(((u128)lo + ((u128)hi << 32)) % GL_P);
    }
}

// ---- device --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mad_i64(int a, int b, u64 c) {          // a * b + c, signed 32 x 32 + 64
    u64 d, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
// eight signed byte columns + the output's two addends -> some u64 congruent to the output
__device__ __forceinline__ u64 recombine(int c0, int c1, int c2, int c3, int c4, int c5, int c6, int c7, const ulonglong2 K, int one, int s16) {
    const int e0 = (c1 << 8) + c0, e1 = (c3 << 8) + c2, e2 = (c5 << 8) + c4, e3 = (c7 << 8) + c6;   // |e| < 2^29
    const u64 lo = mad_i64(e1, s16, mad_i64(e0, one, K.x));            // columns 0..3 + KL: in (2^46, 2^49)
    const u64 hi = mad_i64(e3, s16, mad_i64(e2, one, K.y));            // columns 4..7 + KH, weight 2^32
    const u64 q = add_word(hi, (u32)(lo >> 32));                        // value = lo.lo + 2^32 q,  q < 2^50
    return gl::mad_eps_nc((u32)(q >> 32), gl::mk64((u32)lo, (u32)q));   // 2^64 = 2^32 - 1
}
__device__ __forceinline__ void swap32(u32& a, u32& b) {                // a's lanes 32..63 <-> b's lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap64(u64& a, u64& b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    swap32(a0, b0); swap32(a1, b1);
    a = gl::mk64(a0, a1); b = gl::mk64(b0, b1);
}

// the same + ad (any u64): the addend joins the low 64 bits, its carry the 2^64 word
__device__ __forceinline__ u64 recombine_add(int c0, int c1, int c2, int c3, int c4, int c5, int c6, int c7, const ulonglong2 K, int one, int s16, u64 ad) {
    const int e0 = (c1 << 8) + c0, e1 = (c3 << 8) + c2, e2 = (c5 << 8) + c4, e3 = (c7 << 8) + c6;
    const u64 lo = mad_i64(e1, s16, mad_i64(e0, one, K.x));
    const u64 hi = mad_i64(e3, s16, mad_i64(e2, one, K.y));
    const u64 q = add_word(hi, (u32)(lo >> 32));
    u32 ca, cb, cc;
    const u32 x0 = __builtin_addc((u32)lo, (u32)ad, 0u, &ca);
    const u32 x1 = __builtin_addc((u32)q, (u32)(ad >> 32), ca, &cb);
    const u32 r2 = __builtin_addc((u32)(q >> 32), 0u, cb, &cc);
    return gl::mad_eps_nc(r2, gl::mk64(x0, x1));
}
// The B operands of a product: the wave's 64 vectors as signed bytes, KS k-steps x 2 column tiles x 4 registers.
template <int KS> struct BOpsT { v4i b[KS][2]; };
using BOps = BOpsT<3>;
template <int N_IN, int KS, class X>
__device__ __forceinline__ void make_b(BOpsT<KS>& B, X&& x /* x(j) -> u64, any representative */) {
    static_for<0, KS>([&](auto SI) {
        constexpr int s = decltype(SI)::value;
        u32 va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ja = 4 * s + q, jb = 4 * s + 2 + q;
            const u64 xa = ja < N_IN ? x(ja) : 0, xb = jb < N_IN ? x(jb) : 0;
            va[2 * q] = (u32)xa ^ 0x80808080u; va[2 * q + 1] = (u32)(xa >> 32) ^ 0x80808080u;
            vb[2 * q] = (u32)xb ^ 0x80808080u; vb[2 * q + 1] = (u32)(xb >> 32) ^ 0x80808080u;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) swap32(va[r], vb[r]);   // va: lanes < 32 keep their own first half, lanes >= 32 get the partner's second half ...
        B.b[s][0] = v4i{(int)va[0], (int)va[1], (int)va[2], (int)va[3]};   // column tile 0 = the vectors of lanes 0..31
        B.b[s][1] = v4i{(int)vb[0], (int)vb[1], (int)vb[2], (int)vb[3]};   // column tile 1 = the vectors of lanes 32..63
    });
}
template <int N_IN = 12, class X>
__device__ __forceinline__ void make_b(BOps& B, X&& x) { make_b<N_IN, 3>(B, x); }

// out[o] for this lane's vector, o < 4 * N_TILES (N_TILES <= TL, the table's); tab = the product's table (LDS, or global memory that
// stays in cache).  Every lane of the wave must be here.  add(o): the lane's OWN word o (any u64) to be added to output o, or nothing.
struct NoAdd {};
template <int N_TILES, int KS, int TL, class ADD, class OUT>
__device__ __forceinline__ void product_t(const BOpsT<KS>& B, const u64* __restrict__ tab, ADD&& add, OUT&& out /* out(o, value) */) {
    constexpr bool ADDS = !std::is_same<std::decay_t<ADD>, NoAdd>::value;
    const int lane = threadIdx.x & 63;
    const v4i* __restrict__ fr = reinterpret_cast<const v4i*>(tab) + lane;
    const ulonglong2* __restrict__ Kt = reinterpret_cast<const ulonglong2*>(tab + frag_words(KS, TL)) + 2 * (lane >> 5);
    int one = 1, s16 = 65536;
    asm volatile("" : "+v"(one), "+v"(s16));
    static_for<0, N_TILES>([&](auto TI) {
        constexpr int t = decltype(TI)::value;
        v16i acc0 = {}, acc1 = {};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i a = fr[(t * KS + s) * 64];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B.b[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B.b[s][1], acc1, 0, 0, 0);
        }
        const ulonglong2 K0 = Kt[4 * t], K1 = Kt[4 * t + 1];
        u64 r00, r01, r10, r11;
        if constexpr (ADDS) {
            // the addend travels to the lane that recombines output o of this vector the way the result travels back
            u64 a00 = add(4 * t), a01 = add(4 * t + 1), a10 = add(4 * t + 2), a11 = add(4 * t + 3);
            swap64(a00, a10);                              // a00: word 4 t + 2 H of the vector in column tile 0, a10: of the one in tile 1
            swap64(a01, a11);
            r00 = recombine_add(acc0[0], acc0[1], acc0[2], acc0[3], acc0[4], acc0[5], acc0[6], acc0[7], K0, one, s16, a00);
            r01 = recombine_add(acc0[8], acc0[9], acc0[10], acc0[11], acc0[12], acc0[13], acc0[14], acc0[15], K1, one, s16, a01);
            r10 = recombine_add(acc1[0], acc1[1], acc1[2], acc1[3], acc1[4], acc1[5], acc1[6], acc1[7], K0, one, s16, a10);
            r11 = recombine_add(acc1[8], acc1[9], acc1[10], acc1[11], acc1[12], acc1[13], acc1[14], acc1[15], K1, one, s16, a11);
        } else {
            r00 = recombine(acc0[0], acc0[1], acc0[2], acc0[3], acc0[4], acc0[5], acc0[6], acc0[7], K0, one, s16);
            r01 = recombine(acc0[8], acc0[9], acc0[10], acc0[11], acc0[12], acc0[13], acc0[14], acc0[15], K1, one, s16);
            r10 = recombine(acc1[0], acc1[1], acc1[2], acc1[3], acc1[4], acc1[5], acc1[6], acc1[7], K0, one, s16);
            r11 = recombine(acc1[8], acc1[9], acc1[10], acc1[11], acc1[12], acc1[13], acc1[14], acc1[15], K1, one, s16);
        }
        swap64(r00, r10);                                  // r00: output 4 t of the lane's own vector, r10: output 4 t + 2
        swap64(r01, r11);
        out(4 * t, r00); out(4 * t + 1, r01); out(4 * t + 2, r10); out(4 * t + 3, r11);
    });
}
// the 12 x 12 shapes of Poseidon
template <int N_TILES = 3, class OUT>
__device__ __forceinline__ void product(const BOps& B, const u64* __restrict__ tab, OUT&& out) { product_t<N_TILES, 3, 3>(B, tab, NoAdd{}, out); }
template <int N_TILES = 3, class ADD, class OUT>
__device__ __forceinline__ void product_add(const BOps& B, const u64* __restrict__ tab, ADD&& add, OUT&& out) { product_t<N_TILES, 3, 3>(B, tab, add, out); }

}  // namespace pmfma
}  // namespace zk
