// Prime-field arithmetic in 29-bit limbs for gfx950, shared by the G1 MSM (msm_impl.hip.h: Fq of BN254 and
// BLS12-381) and the BN128-field Poseidon (poseidon_bn128.hip: Fr of BN254).  Included inside a namespace that
// provides NL (32-bit limbs of the external Montgomery form, R = 2^(32 NL)), NR (29-bit limbs, R' = 2^(29 NR)),
// Q29, QINV29, ONE29 (R' mod q), CIN29 (R'^2/R mod q), COUT29 (R mod q), Q2_29 / Q4_29 / Q8_29.
// No include guard on purpose.
//
// gfx950 has no cheap carry chain (every v_addc costs two wait states), but v_mad_u64_u32 adds a full 64-bit
// accumulator for free.  The 2 NR^2 partial products of a Montgomery multiplication (product + reduction)
// accumulate in 64-bit column registers with no carry handling at all (NR * 2^58 * 2 < 2^63), one
// shift-and-add per column at the end.  tools/ubench_fq.hip: 156 G products/s against 80 G for the 8 x 32-bit
// CIOS form.  Values are kept lazily reduced: products are < 2q, sums and differences carry explicit bounds,
// every pair of multiplicands a < Aq, b < Bq must satisfy A*B <= 168 (BN254; far looser for BLS12-381) so that
// ab/R' + q stays < 2q.
constexpr int LB = 29;
constexpr u32 LMASK = (1u << LB) - 1;
struct fe { u32 l[NR]; };

__device__ __forceinline__ fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = 0;
    return r;
}
__device__ __forceinline__ fe fe_one() {
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = ONE29(i);
    return r;
}
// carry propagation: limbs back below 2^29 (the top limb keeps the excess); _s for signed limbs
__device__ __forceinline__ void fe_norm_u(fe& a) {
#pragma unroll
    for (int i = 0; i + 1 < NR; ++i) { a.l[i + 1] += a.l[i] >> LB; a.l[i] &= LMASK; }
}
__device__ __forceinline__ void fe_norm_s(fe& a) {
#pragma unroll
    for (int i = 0; i + 1 < NR; ++i) { a.l[i + 1] += (u32)((int)a.l[i] >> LB); a.l[i] &= LMASK; }
}
__device__ __forceinline__ fe fe_add(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = a.l[i] + b.l[i];
    fe_norm_u(r);
    return r;
}
__device__ __forceinline__ fe fe_dbl(const fe& a) { return fe_add(a, a); }
// a - b + M q for M in {2, 4, 8}; requires b <= M q so that the value stays non-negative
template <int M>
__device__ __forceinline__ fe fe_sub(const fe& a, const fe& b) {
    static_assert(M == 2 || M == 4 || M == 8, "bias");
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = a.l[i] + (M == 2 ? Q2_29(i) : M == 4 ? Q4_29(i) : Q8_29(i)) - b.l[i];
    fe_norm_s(r);
    return r;
}
// Montgomery product a*b/R' mod q: operands with limbs <= 2^29 and values < 11q, result < 2q, limbs normalised
#ifndef FQ_MUL_ATTR
#define FQ_MUL_ATTR __forceinline__
#endif
__device__ FQ_MUL_ATTR fe fe_mul(const fe& a, const fe& b) {
    u64 t[2 * NR];
#pragma unroll
    for (int i = 0; i < 2 * NR; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const u32 m = ((u32)t[i] * QINV29) & LMASK;
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)m * Q29(j);
        t[i + 1] += t[i] >> LB;                       // the low 29 bits of t[i] are zero now
    }
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k + 1 < NR) { r.l[k] = (u32)t[NR + k] & LMASK; t[NR + k + 1] += t[NR + k] >> LB; }
        else r.l[k] = (u32)t[NR + k];
    }
    return r;
}
// (a*b + c*d)/R' mod q with ONE reduction: both sets of partial products accumulate in the same 64-bit columns
// (3 NR * 2^58 < 2^64).  Bounds: A*B + C*D <= 168 (BN254) for a result < 2q.  Half the work of two products and
// a renormalised sum -- the Fq2 product (msm_impl.hip.h) is two of these.
__device__ __forceinline__ fe fe_mul2(const fe& a, const fe& b, const fe& c, const fe& d) {
    u64 t[2 * NR];
#pragma unroll
    for (int i = 0; i < 2 * NR; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)c.l[i] * d.l[j];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const u32 m = ((u32)t[i] * QINV29) & LMASK;
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)m * Q29(j);
        t[i + 1] += t[i] >> LB;
    }
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k + 1 < NR) { r.l[k] = (u32)t[NR + k] & LMASK; t[NR + k + 1] += t[NR + k] >> LB; }
        else r.l[k] = (u32)t[NR + k];
    }
    return r;
}
// Deferred reduction for sums of products: partial products of up to FE_WIDE_MAX pairs accumulate in the 64-bit
// columns ((FE_WIDE_MAX + 1) NR 2^58 <= 2^64 with the reduction's own products), then ONE Montgomery reduction.
// Bounds: sum A_i B_i <= 168 (BN254 fields) / 68 (BLS12-381 Fr) for a result < 2q.  A dot product of n terms costs
// n + ceil(n / 6) column passes instead of 2 n.
constexpr int FE_WIDE_MAX = 6;
struct fe_wide { u64 t[2 * NR]; };
__device__ __forceinline__ void fe_wide_zero(fe_wide& w) {
#pragma unroll
    for (int i = 0; i < 2 * NR; ++i) w.t[i] = 0;
}
__device__ __forceinline__ void fe_wide_mac(fe_wide& w, const fe& a, const fe& b) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j < NR; ++j) w.t[i + j] += (u64)a.l[i] * b.l[j];
    }
}
__device__ __forceinline__ fe fe_wide_reduce(fe_wide& w) {   // consumes w
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const u32 m = ((u32)w.t[i] * QINV29) & LMASK;
#pragma unroll
        for (int j = 0; j < NR; ++j) w.t[i + j] += (u64)m * Q29(j);
        w.t[i + 1] += w.t[i] >> LB;
    }
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k + 1 < NR) { r.l[k] = (u32)w.t[NR + k] & LMASK; w.t[NR + k + 1] += w.t[NR + k] >> LB; }
        else r.l[k] = (u32)w.t[NR + k];
    }
    return r;
}
// a^2: the NR (NR - 1) / 2 cross products once, against a doubled limb (same column sums as fe_mul(a, a), 36 multiply-adds fewer)
#ifdef ZK_FE_SQR_PLAIN   // variant switch for A/B runs
__device__ __forceinline__ fe fe_sqr(const fe& a) { return fe_mul(a, a); }
#else
__device__ FQ_MUL_ATTR fe fe_sqr(const fe& a) {
    u64 t[2 * NR];
#pragma unroll
    for (int i = 0; i < 2 * NR; ++i) t[i] = 0;
    u32 d[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) d[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        t[2 * i] += (u64)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < NR; ++j) t[i + j] += (u64)d[i] * a.l[j];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const u32 m = ((u32)t[i] * QINV29) & LMASK;
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)m * Q29(j);
        t[i + 1] += t[i] >> LB;
    }
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k + 1 < NR) { r.l[k] = (u32)t[NR + k] & LMASK; t[NR + k + 1] += t[NR + k] >> LB; }
        else r.l[k] = (u32)t[NR + k];
    }
    return r;
}
#endif
// c + a*b/R' mod q in one reduction: c's limbs start the high columns (c R' + a b, then / R').  Result < c + ab/R' + q: the
// caller carries the bound (it grows by about q per call) and renormalises; limbs of b may reach 2^30 (9 x 2^59 + the
// reduction's 9 x 2^58 stay below 2^63).
__device__ FQ_MUL_ATTR fe fe_mul_acc(const fe& a, const fe& b, const fe& c) {
    u64 t[2 * NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { t[i] = 0; t[NR + i] = c.l[i]; }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const u32 m = ((u32)t[i] * QINV29) & LMASK;
#pragma unroll
        for (int j = 0; j < NR; ++j) t[i + j] += (u64)m * Q29(j);
        t[i + 1] += t[i] >> LB;
    }
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k + 1 < NR) { r.l[k] = (u32)t[NR + k] & LMASK; t[NR + k + 1] += t[NR + k] >> LB; }
        else r.l[k] = (u32)t[NR + k];
    }
    return r;
}
// x == 0 (mod q) for a product x (< 2q, normalised): x is 0 or q
__device__ __forceinline__ bool fe_is_zero_m(const fe& a) {
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) { z |= a.l[i]; e |= a.l[i] ^ Q29(i); }
    return z == 0 || e == 0;
}
// canonical representative of a value < 2q
__device__ __forceinline__ fe fe_canon(const fe& a) {
    fe t;
#pragma unroll
    for (int i = 0; i < NR; ++i) t.l[i] = a.l[i] - Q29(i);
    fe_norm_s(t);
    const bool neg = (int)t.l[NR - 1] < 0;
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = neg ? a.l[i] : t.l[i];
    return r;
}
// a^(q - 2).  (Round 6 measured the binary extended Euclid in its place -- at most 2 log2 q halvings and subtractions of ~100 dependent
// instructions against ~1.5 log2 q products of 225 / 500 instructions -- and lost: msm_final_kernel 543 -> 677 us, msm_final_bits_kernel
// 1192 -> 1225 us, g1_mul_generator_kernel 42.7 -> 63.8 ms.  A product's 2 x 81 multiply-adds are independent and issue every 4.2 cycles even
// on a lone wave; Euclid's compare / borrow chains are dependent and pay the lone wave's 8.3 cycles each.  tools/experiments/fe_inv_euclid.patch)
__host__ __device__ constexpr u32 fe_qm2_limb(int i) {   // limb i of q - 2 (the borrow may run past limb 0: BLS12-381's r = 1 mod 2^29)
    long long borrow = 2;
    u32 out = 0;
    for (int k = 0; k <= i; ++k) {
        long long v = (long long)Q29(k) - borrow;
        borrow = 0;
        if (v < 0) { v += (1ll << LB); borrow = 1; }
        out = (u32)v;
    }
    return out;
}
__device__ fe fe_inv(const fe& a) {  // a^(q-2)
    fe r = fe_one();
    for (int i = NR - 1; i >= 0; --i) {
        const u32 w = fe_qm2_limb(i);
        for (int b = LB - 1; b >= 0; --b) {
            r = fe_sqr(r);
            if ((w >> b) & 1) r = fe_mul(r, a);
        }
    }
    return r;
}
// external layout (NL x 32-bit words, Montgomery R) <-> internal (NR x 29-bit limbs, Montgomery R')
__device__ __forceinline__ fe fe_from_std(const u32 (&w)[NL]) {
    fe x;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int bit = LB * k, wi = bit >> 5, s = bit & 31;
        u32 v = wi < NL ? w[wi] >> s : 0;
        if (s > 32 - LB && wi + 1 < NL) v |= w[wi + 1] << (32 - s);
        x.l[k] = v & LMASK;
    }
    fe c;
#pragma unroll
    for (int i = 0; i < NR; ++i) c.l[i] = CIN29(i);
    return fe_mul(x, c);                                // x R * (R'^2/R) / R' = x R'
}
__device__ __forceinline__ void fe_to_std(const fe& a, u32 (&w)[NL]) {
    fe c;
#pragma unroll
    for (int i = 0; i < NR; ++i) c.l[i] = COUT29(i);
    const fe x = fe_canon(fe_mul(a, c));                 // x R' * R / R' = x R, canonical
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int bit = 32 * j, k = bit / LB, s = bit % LB;
        u32 v = x.l[k] >> s;
        if (k + 1 < NR) v |= x.l[k + 1] << (LB - s);
        if (k + 2 < NR && 2 * LB - s < 32) v |= x.l[k + 2] << (2 * LB - s);
        w[j] = v;
    }
}

