// Curve-generic body of the G1 MSM (see msm.hip): included once per curve inside that curve's namespace,
// which provides
//   NL            32-bit limbs of Fq in the external (bellman) layout, Montgomery R = 2^(32 NL)
//   NR            29-bit limbs of the internal representation, Montgomery R' = 2^(29 NR)
//   Q29, QINV29, ONE29 (R' mod q), CIN29 (R'^2/R mod q), COUT29 (R mod q), Q2_29/Q4_29/Q8_29 (2q, 4q, 8q)
//   GEN_X, GEN_Y  the generator in the external layout
// The curve is y^2 = x^3 + b with a = 0 (the formulas never touch b).  No include guard on purpose.
//
// Field arithmetic: fe29_impl.hip.h (29-bit limbs, 64-bit column accumulators, lazily reduced values).
#include "fe29_impl.hip.h"

// ---- coordinate field `cf`: Fq for G1; Fq2 = Fq[u]/(u^2 + 1) for G2 (MSM_G2), built from the Fq operations
// with every component brought back below 2q after a product (one extra product with R' mod q), so that the
// bounds of the point formulas below hold for both.
#ifndef MSM_G2
typedef fe cf;
constexpr int CW_STD = NL, CW_INT = NR;   // words per coordinate: external layout / internal limbs
__device__ __forceinline__ cf cf_zero() { return fe_zero(); }
__device__ __forceinline__ cf cf_one() { return fe_one(); }
__device__ __forceinline__ cf cf_add(const cf& a, const cf& b) { return fe_add(a, b); }
__device__ __forceinline__ cf cf_dbl(const cf& a) { return fe_dbl(a); }
template <int M> __device__ __forceinline__ cf cf_sub(const cf& a, const cf& b) { return fe_sub<M>(a, b); }
__device__ __forceinline__ cf cf_mul(const cf& a, const cf& b) { return fe_mul(a, b); }
__device__ __forceinline__ cf cf_sqr(const cf& a) { return fe_sqr(a); }
__device__ __forceinline__ bool cf_is_zero_m(const cf& a) { return fe_is_zero_m(a); }
__device__ __forceinline__ cf cf_inv(const cf& a) { return fe_inv(a); }
__device__ __forceinline__ cf cf_from_std(const u32* w) {
    u32 t[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) t[i] = w[i];
    return fe_from_std(t);
}
__device__ __forceinline__ void cf_to_std(const cf& a, u32* w) {
    u32 t[NL];
    fe_to_std(a, t);
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = t[i];
}
__device__ __forceinline__ cf cf_load_int(const u32* p) {
    cf a;
#pragma unroll
    for (int k = 0; k < NR; ++k) a.l[k] = p[k];
    return a;
}
__device__ __forceinline__ void cf_store_int(const cf& a, u32* p) {
#pragma unroll
    for (int k = 0; k < NR; ++k) p[k] = a.l[k];
}
#else
struct cf { fe c0, c1; };                 // c0 + c1 u, u^2 = -1 (both BN254 and BLS12-381 build Fq2 this way)
constexpr int CW_STD = 2 * NL, CW_INT = 2 * NR;
__device__ __forceinline__ fe fe_renorm(const fe& a) { return fe_mul(a, fe_one()); }   // < 168q -> < 2q, same residue
__device__ __forceinline__ cf cf_zero() { cf r; r.c0 = fe_zero(); r.c1 = fe_zero(); return r; }
__device__ __forceinline__ cf cf_one() { cf r; r.c0 = fe_one(); r.c1 = fe_zero(); return r; }
__device__ __forceinline__ cf cf_add(const cf& a, const cf& b) { cf r; r.c0 = fe_add(a.c0, b.c0); r.c1 = fe_add(a.c1, b.c1); return r; }
__device__ __forceinline__ cf cf_dbl(const cf& a) { return cf_add(a, a); }
template <int M> __device__ __forceinline__ cf cf_sub(const cf& a, const cf& b) { cf r; r.c0 = fe_sub<M>(a.c0, b.c0); r.c1 = fe_sub<M>(a.c1, b.c1); return r; }
// (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u with one reduction per component (fe_mul2): the subtrahend enters as
// a1 (8q - b1).  Requires b.c1 <= 8q and, for components a < Aq, b < Bq, A (B + 8) <= 168 (BN254; every call site
// below keeps the operand with the larger bound first: the worst is 10q x 6q = 140).  Components of the result < 2q.
#undef CF_MUL_ATTR
#undef PT_COLD_ATTR
#ifdef MSM_G2_INLINE_CF
#define CF_MUL_ATTR __forceinline__
#define PT_COLD_ATTR __noinline__
#else
#define CF_MUL_ATTR __noinline__
#define PT_COLD_ATTR
#endif
__device__ CF_MUL_ATTR cf cf_mul(const cf& a, const cf& b) {
    cf r;
    r.c0 = fe_mul2(a.c0, b.c0, a.c1, fe_sub<8>(fe_zero(), b.c1));
    r.c1 = fe_mul2(a.c0, b.c1, a.c1, b.c0);
    return r;
}
__device__ CF_MUL_ATTR cf cf_sqr(const cf& a) {   // a <= 8q (or c0 < 10q with c1 < 2q): a0^2 + a1 (8q - a1), 2 a0 a1
    cf r;
    r.c0 = fe_mul2(a.c0, a.c0, a.c1, fe_sub<8>(fe_zero(), a.c1));
    r.c1 = fe_mul(fe_dbl(a.c0), a.c1);
    return r;
}
__device__ __forceinline__ bool cf_is_zero_m(const cf& a) { return fe_is_zero_m(a.c0) && fe_is_zero_m(a.c1); }
__device__ cf cf_inv(const cf& a) {       // (a0 - a1 u) / (a0^2 + a1^2), a < 2q
    const fe n = fe_inv(fe_renorm(fe_add(fe_sqr(a.c0), fe_sqr(a.c1))));
    cf r; r.c0 = fe_mul(a.c0, n); r.c1 = fe_mul(fe_sub<2>(fe_zero(), a.c1), n);
    return r;
}
__device__ __forceinline__ cf cf_from_std(const u32* w) {
    u32 t0[NL], t1[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { t0[i] = w[i]; t1[i] = w[NL + i]; }
    cf r; r.c0 = fe_from_std(t0); r.c1 = fe_from_std(t1);
    return r;
}
__device__ __forceinline__ void cf_to_std(const cf& a, u32* w) {
    u32 t0[NL], t1[NL];
    fe_to_std(a.c0, t0); fe_to_std(a.c1, t1);
#pragma unroll
    for (int i = 0; i < NL; ++i) { w[i] = t0[i]; w[NL + i] = t1[i]; }
}
__device__ __forceinline__ cf cf_load_int(const u32* p) {
    cf a;
#pragma unroll
    for (int k = 0; k < NR; ++k) { a.c0.l[k] = p[k]; a.c1.l[k] = p[NR + k]; }
    return a;
}
__device__ __forceinline__ void cf_store_int(const cf& a, u32* p) {
#pragma unroll
    for (int k = 0; k < NR; ++k) { p[k] = a.c0.l[k]; p[NR + k] = a.c1.l[k]; }
}
#endif

// XYZZ coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0.
// Invariants of every stored point: X < 8q, Y <= 4q, ZZ, ZZZ < 2q (products), limbs normalised.
struct xyzz { cf X, Y, ZZ, ZZZ; };
struct aff { cf x, y; };  // x, y < 2q

__device__ __forceinline__ xyzz pt_inf() {
    xyzz p; p.X = cf_zero(); p.Y = cf_zero(); p.ZZ = cf_zero(); p.ZZZ = cf_zero();
    return p;
}
__device__ __forceinline__ bool pt_is_inf(const xyzz& p) { return cf_is_zero_m(p.ZZ); }
// shared tail of the addition formulas: given U1 (= X1 scaled, < 8q), S1 (<= 4q), P, R and PP = P^2
__device__ __forceinline__ void pt_finish(xyzz& r, const cf& U1, const cf& S1, const cf& P, const cf& Rr, const cf& PP) {
    const cf PPP = cf_mul(P, PP), Q = cf_mul(U1, PP);                      // < 2q each
    r.X = cf_sub<4>(cf_sub<2>(cf_sqr(Rr), PPP), cf_dbl(Q));                 // < 2q + 2q + 4q = 8q
#ifdef MSM_G2
    r.Y = cf_sub<2>(cf_mul(cf_sub<8>(Q, r.X), Rr), cf_mul(S1, PPP));        // (Q - X3 < 10q) first, Rr < 6q ; Y3 < 4q
#else
    // round 6: (Q - X3) R - S1 PPP as ONE sum of two products with one Montgomery reduction (fe_mul2; bounds 10 * 6 + 4 * 2 = 68 <= 168):
    // a reduction less per point addition, 110 of ~2 700 instructions.  Y3 < 2q.
    r.Y = fe_mul2(cf_sub<8>(Q, r.X), Rr, cf_sub<4>(cf_zero(), S1), PPP);
#endif
}
#ifndef PT_COLD_ATTR
#define PT_COLD_ATTR
#endif
__device__ PT_COLD_ATTR xyzz pt_dbl_aff(const aff& a) {  // mdbl-2008-s-1 (a = 0)
    const cf U = cf_dbl(a.y), V = cf_sqr(U), W = cf_mul(U, V), S = cf_mul(a.x, V);
    const cf xx = cf_sqr(a.x), M = cf_add(cf_dbl(xx), xx);                  // < 6q
    xyzz r;
    r.X = cf_sub<4>(cf_sqr(M), cf_dbl(S));                                  // < 6q
    r.Y = cf_sub<2>(cf_mul(cf_sub<8>(S, r.X), M), cf_mul(W, a.y));
    r.ZZ = V; r.ZZZ = W;
    return r;
}
__device__ PT_COLD_ATTR xyzz pt_dbl(const xyzz& p) {  // dbl-2008-s-1 (a = 0)
    if (pt_is_inf(p)) return p;
    const cf U = cf_dbl(p.Y), V = cf_sqr(U), W = cf_mul(U, V), S = cf_mul(p.X, V);   // U <= 8q
    const cf xx = cf_sqr(p.X), M = cf_add(cf_dbl(xx), xx);                  // < 6q
    xyzz r;
    r.X = cf_sub<4>(cf_sqr(M), cf_dbl(S));                                  // < 6q
    r.Y = cf_sub<2>(cf_mul(cf_sub<8>(S, r.X), M), cf_mul(W, p.Y));
    r.ZZ = cf_mul(V, p.ZZ); r.ZZZ = cf_mul(W, p.ZZZ);
    return r;
}
__device__ __forceinline__ xyzz pt_madd(const xyzz& p, const aff& a) {  // madd-2008-s
    if (pt_is_inf(p)) { xyzz r; r.X = a.x; r.Y = a.y; r.ZZ = cf_one(); r.ZZZ = cf_one(); return r; }
    const cf U2 = cf_mul(a.x, p.ZZ), S2 = cf_mul(a.y, p.ZZZ);
    cf P = cf_sub<8>(U2, p.X);                                              // < 10q
    const cf Rr = cf_sub<4>(S2, p.Y);                                       // < 6q
#ifdef MSM_G2
    P.c1 = fe_renorm(P.c1);                                                 // cf_sqr's bound: c0 < 10q needs c1 < 2q
#endif
    const cf PP = cf_sqr(P);
    if (cf_is_zero_m(PP)) return cf_is_zero_m(cf_sqr(Rr)) ? pt_dbl_aff(a) : pt_inf();  // q prime: P^2 = 0 <=> P = 0
    xyzz r;
    pt_finish(r, p.X, p.Y, P, Rr, PP);
    r.ZZ = cf_mul(p.ZZ, PP); r.ZZZ = cf_mul(p.ZZZ, cf_mul(P, PP));
    return r;
}
__device__ PT_COLD_ATTR xyzz pt_add(const xyzz& p, const xyzz& q) {  // add-2008-s
    if (pt_is_inf(p)) return q;
    if (pt_is_inf(q)) return p;
    const cf U1 = cf_mul(p.X, q.ZZ), U2 = cf_mul(q.X, p.ZZ), S1 = cf_mul(p.Y, q.ZZZ), S2 = cf_mul(q.Y, p.ZZZ);
    const cf P = cf_sub<2>(U2, U1), Rr = cf_sub<2>(S2, S1);                 // < 4q
    const cf PP = cf_sqr(P);
    if (cf_is_zero_m(PP)) return cf_is_zero_m(cf_sqr(Rr)) ? pt_dbl(p) : pt_inf();
    xyzz r;
    pt_finish(r, U1, S1, P, Rr, PP);
    r.ZZ = cf_mul(cf_mul(p.ZZ, q.ZZ), PP); r.ZZZ = cf_mul(cf_mul(p.ZZZ, q.ZZZ), cf_mul(P, PP));
    return r;
}
__device__ __forceinline__ xyzz pt_neg(const xyzz& p) {
    xyzz r = p;
    r.Y = cf_sub<4>(cf_zero(), p.Y);                                        // 4q - Y <= 4q
    return r;
}
// ---- four lanes per point operation -------------------------------------------------------------------------------------------
// The tails of a sum -- the bucket hierarchy and the final walk over the windows -- are chains of a few hundred dependent point
// operations on a handful of lanes: latency, not throughput.  A doubling is 9 field products in 3 dependent stages, a full addition
// 14 in 4; here the lanes of a quad hold the same point and lane q computes the q-th product of each stage, the results travel by
// DPP quad broadcasts (a move per limb, no LDS).  The same formulas, the same operands, the same bounds: bit-identical results, in
// a third of the dependent products.  Called by all four lanes of a quad with identical arguments.
__device__ __forceinline__ u32 quad_word(u32 v, int k) {            // lane k's value in every lane of the quad (k is uniform)
    switch (k) {
        case 0: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false);
        case 1: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, false);
        case 2: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, false);
        default: return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, false);
    }
}
__device__ __forceinline__ fe quad_fe(const fe& v, int k) {
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = quad_word(v.l[i], k);
    return r;
}
__device__ __forceinline__ fe pick_fe(int q, const fe& a0, const fe& a1, const fe& a2, const fe& a3) {
    // by masks, not by `?:` -- the compiler turns a select between structures into four branches under EXEC, each with its own
    // copy of the product that follows, and the four lanes then take their turns
    u32 m0 = q == 0 ? ~0u : 0u, m1 = q == 1 ? ~0u : 0u, m2 = q == 2 ? ~0u : 0u, m3 = q == 3 ? ~0u : 0u;
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
    fe r;
#pragma unroll
    for (int i = 0; i < NR; ++i) r.l[i] = (a0.l[i] & m0) | (a1.l[i] & m1) | (a2.l[i] & m2) | (a3.l[i] & m3);
    return r;
}
__device__ __forceinline__ void fe_opaque(fe& v) {
#pragma unroll
    for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(v.l[i]));
}
#ifndef MSM_G2
__device__ __forceinline__ void cf_opaque(cf& v) { fe_opaque(v); }
__device__ __forceinline__ cf quad_cf(const cf& v, int k) { return quad_fe(v, k); }
__device__ __forceinline__ cf pick_cf(int q, const cf& a0, const cf& a1, const cf& a2, const cf& a3) { return pick_fe(q, a0, a1, a2, a3); }
#else
__device__ __forceinline__ void cf_opaque(cf& v) { fe_opaque(v.c0); fe_opaque(v.c1); }
__device__ __forceinline__ cf quad_cf(const cf& v, int k) { cf r; r.c0 = quad_fe(v.c0, k); r.c1 = quad_fe(v.c1, k); return r; }
__device__ __forceinline__ cf pick_cf(int q, const cf& a0, const cf& a1, const cf& a2, const cf& a3) {
    cf r; r.c0 = pick_fe(q, a0.c0, a1.c0, a2.c0, a3.c0); r.c1 = pick_fe(q, a0.c1, a1.c1, a2.c1, a3.c1); return r;
}
#endif
// one stage: lane q multiplies (a_q, b_q); out[k] = the product of lane k, in every lane
__device__ PT_COLD_ATTR void quad_stage(const cf& a0, const cf& b0, const cf& a1, const cf& b1, const cf& a2, const cf& b2, const cf& a3, const cf& b3,
                                        cf& o0, cf& o1, cf& o2, cf& o3) {
    const int q = threadIdx.x & 3;
    cf prod = cf_mul(pick_cf(q, a0, a1, a2, a3), pick_cf(q, b0, b1, b2, b3));
    cf_opaque(prod);                                                       // (the broadcasts read exactly this value)
    o0 = quad_cf(prod, 0); o1 = quad_cf(prod, 1); o2 = quad_cf(prod, 2); o3 = quad_cf(prod, 3);
    cf_opaque(o0); cf_opaque(o1); cf_opaque(o2); cf_opaque(o3);
}
__device__ PT_COLD_ATTR xyzz pt_dbl4(const xyzz& p) {   // pt_dbl, three stages
    if (pt_is_inf(p)) return p;
    const cf U = cf_dbl(p.Y);
    cf V, xx, W, S, MM, t1, t2, d0, d1;
    quad_stage(U, U, p.X, p.X, U, U, U, U, V, xx, d0, d1);                      // V = U^2, xx = X^2
    const cf M = cf_add(cf_dbl(xx), xx);
    quad_stage(U, V, p.X, V, M, M, U, V, W, S, MM, d0);                          // W = U V, S = X V, M^2
    xyzz r;
    r.X = cf_sub<4>(MM, cf_dbl(S));
    quad_stage(cf_sub<8>(S, r.X), M, W, p.Y, V, p.ZZ, W, p.ZZZ, t1, t2, r.ZZ, r.ZZZ);
    r.Y = cf_sub<2>(t1, t2);
    return r;
}
__device__ PT_COLD_ATTR xyzz pt_add4(const xyzz& p, const xyzz& q) {   // pt_add, four stages
    if (pt_is_inf(p)) return q;
    if (pt_is_inf(q)) return p;
    cf U1, U2, S1, S2, PP, RR, ZZ12, ZZZ12, PPP, Q, t1, t2, d0;
    quad_stage(p.X, q.ZZ, q.X, p.ZZ, p.Y, q.ZZZ, q.Y, p.ZZZ, U1, U2, S1, S2);
    const cf P = cf_sub<2>(U2, U1), Rr = cf_sub<2>(S2, S1);                      // < 4q
    quad_stage(P, P, Rr, Rr, p.ZZ, q.ZZ, p.ZZZ, q.ZZZ, PP, RR, ZZ12, ZZZ12);
    if (cf_is_zero_m(PP)) return cf_is_zero_m(RR) ? pt_dbl4(p) : pt_inf();
    xyzz r;
    quad_stage(P, PP, U1, PP, ZZ12, PP, P, PP, PPP, Q, r.ZZ, d0);
    r.X = cf_sub<4>(cf_sub<2>(RR, PPP), cf_dbl(Q));
    quad_stage(cf_sub<8>(Q, r.X), Rr, S1, PPP, ZZZ12, PPP, S1, PPP, t1, t2, r.ZZZ, d0);
    r.Y = cf_sub<2>(t1, t2);
    return r;
}
// affine (external layout) of a finite point: x = X/ZZ, y = Y/ZZZ; 1/ZZ = (ZZ/ZZZ)^2 because ZZ^3 = ZZZ^2
__device__ void pt_to_std(const xyzz& p, u32* x, u32* y) {   // CW_STD words each
    const cf izzz = cf_inv(p.ZZZ), t = cf_mul(p.ZZ, izzz), izz = cf_sqr(t);
    cf_to_std(cf_mul(p.X, izz), x); cf_to_std(cf_mul(p.Y, izzz), y);
}

#ifndef MSM_N_WIN
#define MSM_N_WIN 16      /* 254-bit scalars: 16 windows of 16 bits */
#endif
#ifndef MSM_SC_WORDS
#define MSM_SC_WORDS 8    /* 32-bit words between two scalars of the input array */
#endif
constexpr int C_BITS = 16, N_WIN = MSM_N_WIN, N_BUCKET = 1 << C_BITS;

__global__ __launch_bounds__(256) void msm_count_kernel(const u32* __restrict__ scalars, u64 n, u32* __restrict__ counts) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;  // one lane per (point, window)
    if (t >= n * N_WIN) return;
    const u64 i = t / N_WIN; const u32 w = t % N_WIN;
    const u32 word = scalars[i * MSM_SC_WORDS + (w >> 1)];
    const u32 d = (w & 1) ? word >> 16 : word & 0xFFFF;
    if (d) atomicAdd(&counts[w * N_BUCKET + d], 1u);
}
__global__ __launch_bounds__(256) void msm_scatter_kernel(const u32* __restrict__ scalars, u64 n, const u32* __restrict__ offsets,
                                   u32* __restrict__ cursors, u32* __restrict__ idx) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N_WIN) return;
    const u64 i = t / N_WIN; const u32 w = t % N_WIN;
    const u32 word = scalars[i * MSM_SC_WORDS + (w >> 1)];
    const u32 d = (w & 1) ? word >> 16 : word & 0xFFFF;
    if (!d) return;
    const u32 key = w * N_BUCKET + d;
    idx[offsets[key] + atomicAdd(&cursors[key], 1u)] = (u32)i;
}
// exclusive scan of 2^20 counters, 1024 per block
__global__ __launch_bounds__(256) void scan_block_kernel(const u32* __restrict__ in, u32* __restrict__ out, u32* __restrict__ block_sum) {
    __shared__ u32 lds[256];
    const u32 base = blockIdx.x * 1024 + threadIdx.x * 4;
    u32 v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = in[base + k]; s += v[k]; }
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        u32 cur = lds[threadIdx.x], prev = threadIdx.x >= (u32)off ? lds[threadIdx.x - off] : 0;
        __syncthreads();
        lds[threadIdx.x] = cur + prev;
        __syncthreads();
    }
    u32 ex = threadIdx.x ? lds[threadIdx.x - 1] : 0;
    if (threadIdx.x == 255) block_sum[blockIdx.x] = lds[255];
#pragma unroll
    for (int k = 0; k < 4; ++k) { out[base + k] = ex; ex += v[k]; }
}
__global__ __launch_bounds__(64) void scan_tops_kernel(u32* __restrict__ block_sum, u32 nb) {  // a few thousand entries: one wave, a chunk per lane
    if (blockIdx.x) return;
    const u32 per = (nb + 63) / 64, lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    u32 sum = 0;
    for (u32 b = lo; b < hi; ++b) sum += block_sum[b];
    u32 incl = sum;                                     // inclusive scan of the chunk sums across the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d, 64); if ((int)threadIdx.x >= d) incl += t; }
    u32 acc = incl - sum;
    for (u32 b = lo; b < hi; ++b) { const u32 t = block_sum[b]; block_sum[b] = acc; acc += t; }
}
__global__ __launch_bounds__(256) void scan_add_kernel(u32* __restrict__ out, const u32* __restrict__ block_sum) {
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 0] += block_sum[blockIdx.x];
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 1] += block_sum[blockIdx.x];
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 2] += block_sum[blockIdx.x];
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 3] += block_sum[blockIdx.x];
}


// ---- bucket sort without global atomics (n < 2^24) ------------------------------------------------------
// key(i, w) = w * 2^16 + digit.  Two LDS-histogram partition passes replace 2 x 16 n device-scope atomics
// on 2^20 counters (2.6 + 3.4 ms at n = 2^22): pass 1 splits the (point, window) pairs into the 4096 coarse
// bins key >> 8 (block histograms [bin][block], one exclusive scan, LDS cursors), pass 2 lets one block per
// coarse bin split its ~16 n / 4096 entries into the 256 fine buckets and emits counts / offsets / idx in
// the layout the accumulation kernel reads.
constexpr int SORT_PTS = 8192;        // points per block in pass 1
constexpr int N_COARSE = N_WIN * 256;  // 4096
__device__ __forceinline__ u32 digit_of(const u32* __restrict__ scalars, u64 i, u32 w) {
    const u32 word = scalars[i * MSM_SC_WORDS + (w >> 1)];
    return (w & 1) ? word >> 16 : word & 0xFFFF;
}
__global__ __launch_bounds__(256) void sort_hist_kernel(const u32* __restrict__ scalars, u64 n, u32 n_blocks, u32* __restrict__ hist /* [N_COARSE][n_blocks] */) {
    __shared__ u32 h[N_COARSE];
    for (int k = threadIdx.x; k < N_COARSE; k += 256) h[k] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * SORT_PTS;
    for (u32 t = threadIdx.x; t < SORT_PTS; t += 256) {   // one point per trip: its 16 digits are independent work
        const u64 i = base + t;
        if (i >= n) break;
        const uint4 lo = reinterpret_cast<const uint4*>(scalars)[(MSM_SC_WORDS / 4) * i];
        const uint4 hi = MSM_SC_WORDS > 4 ? reinterpret_cast<const uint4*>(scalars)[(MSM_SC_WORDS / 4) * i + 1] : uint4{0, 0, 0, 0};
        const u32 wd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int w = 0; w < N_WIN; ++w) {
            const u32 d = (w & 1) ? wd[w >> 1] >> 16 : wd[w >> 1] & 0xFFFF;
            if (d) atomicAdd(&h[w * 256 + (d >> 8)], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N_COARSE; k += 256) hist[(u64)k * n_blocks + blockIdx.x] = h[k];
}
__global__ __launch_bounds__(256) void sort_coarse_kernel(const u32* __restrict__ scalars, u64 n, u32 n_blocks, const u32* __restrict__ hist_scanned,
                                                          u32* __restrict__ coarse /* (i << 8) | low digit byte */) {
    __shared__ u32 cur[N_COARSE];
    for (int k = threadIdx.x; k < N_COARSE; k += 256) cur[k] = hist_scanned[(u64)k * n_blocks + blockIdx.x];
    __syncthreads();
    const u64 base = (u64)blockIdx.x * SORT_PTS;
    for (u32 t = threadIdx.x; t < SORT_PTS; t += 256) {
        const u64 i = base + t;
        if (i >= n) break;
        const uint4 lo = reinterpret_cast<const uint4*>(scalars)[(MSM_SC_WORDS / 4) * i];
        const uint4 hi = MSM_SC_WORDS > 4 ? reinterpret_cast<const uint4*>(scalars)[(MSM_SC_WORDS / 4) * i + 1] : uint4{0, 0, 0, 0};
        const u32 wd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        u32 pos[N_WIN];
#pragma unroll
        for (int w = 0; w < N_WIN; ++w) {
            const u32 d = (w & 1) ? wd[w >> 1] >> 16 : wd[w >> 1] & 0xFFFF;
            pos[w] = d ? atomicAdd(&cur[w * 256 + (d >> 8)], 1u) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int w = 0; w < N_WIN; ++w) {
            const u32 d = (w & 1) ? wd[w >> 1] >> 16 : wd[w >> 1] & 0xFFFF;
            if (d) coarse[pos[w]] = ((u32)i << 8) | (d & 0xFF);
        }
    }
}
// grand total of non-zero (point, window) pairs = last scanned entry + last count
__global__ __launch_bounds__(64) void sort_total_kernel(const u32* __restrict__ hist, const u32* __restrict__ hist_scanned, u32 last, u32* __restrict__ total) {
    if ((threadIdx.x | blockIdx.x) == 0) total[0] = hist_scanned[last] + hist[last];
}
// idx entries: the point's index, or -- for a window table (msm_fixed_*) -- the index of 2^(16 w) P_i in it
__global__ __launch_bounds__(256) void sort_fine_kernel2(const u32* __restrict__ coarse, const u32* __restrict__ hist_scanned, u32 n_blocks, const u32* __restrict__ total,
                                                         u32* __restrict__ counts, u32* __restrict__ offsets, u32* __restrict__ idx, u32 win_stride, u32 base_off) {
    __shared__ u32 h[256], cur[256];
    const u32 bin = blockIdx.x;
    const u32 start = hist_scanned[(u64)bin * n_blocks];
    const u32 end = bin + 1 < (u32)N_COARSE ? hist_scanned[(u64)(bin + 1) * n_blocks] : total[0];
    h[threadIdx.x] = 0;
    __syncthreads();
    // four loads in flight per lane: one load per trip left the pass latency-bound at ~1 TB/s
    for (u32 k = start + threadIdx.x; k < end; k += 1024) {
        u32 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = k + 256 * u < end ? coarse[k + 256 * u] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k + 256 * u < end) atomicAdd(&h[e[u] & 0xFF], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // 256-entry exclusive scan
        u32 acc = start;
        for (int k = 0; k < 256; ++k) { cur[k] = acc; acc += h[k]; }
    }
    __syncthreads();
    counts[bin * 256 + threadIdx.x] = h[threadIdx.x];
    offsets[bin * 256 + threadIdx.x] = cur[threadIdx.x];
    __syncthreads();
    const u32 add = base_off + (bin >> 8) * win_stride;
    for (u32 k = start + threadIdx.x; k < end; k += 1024) {
        u32 e[4], pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = k + 256 * u < end ? coarse[k + 256 * u] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 4; ++u) pos[u] = k + 256 * u < end ? atomicAdd(&cur[e[u] & 0xFF], 1u) : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k + 256 * u < end) idx[pos[u]] = (e[u] >> 8) + add;
    }
}

// ---- bases: external 2*NL words per point -> internal 2*NR limbs, padded to PTW words for 16-byte loads
constexpr int PTW = (2 * CW_INT + 3) / 4 * 4;
__global__ __launch_bounds__(256) void msm_convert_kernel(const u32* __restrict__ bases, u64 n, u32* __restrict__ conv) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[2 * CW_STD];
    const uint4* p = (const uint4*)(bases + i * (2 * CW_STD));
#pragma unroll
    for (int k = 0; k < 2 * CW_STD / 4; ++k) { const uint4 v = p[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
    const cf x = cf_from_std(w), y = cf_from_std(w + CW_STD);
    u32* o = conv + i * PTW;
    cf_store_int(x, o); cf_store_int(y, o + CW_INT);
}
__device__ __forceinline__ aff load_aff(const u32* __restrict__ conv, u32 i) {
    u32 w[PTW];
    const uint4* p = (const uint4*)(conv + (u64)i * PTW);
#pragma unroll
    for (int k = 0; k < PTW / 4; ++k) { const uint4 v = p[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
    aff a;
    a.x = cf_load_int(w); a.y = cf_load_int(w + CW_INT);
    return a;
}
// Load balance: a wave runs as long as its fullest bucket, and bucket sizes are Poisson(n / 2^16).  Buckets are
// therefore handed to lanes in order of decreasing size (a counting sort of the 2^20 bucket ids by size, sizes
// clamped to 1023): the 64 buckets of a wave then differ by at most one point, and the heaviest waves start first.
constexpr int BAL_BINS = 1024;
__global__ __launch_bounds__(256) void balance_hist_kernel(const u32* __restrict__ counts, u32* __restrict__ hist) {
    __shared__ u32 h[BAL_BINS];
    for (int k = threadIdx.x; k < BAL_BINS; k += 256) h[k] = 0;
    __syncthreads();
    const u32 key = blockIdx.x * 256 + threadIdx.x;
    const u32 c = counts[key] < BAL_BINS ? counts[key] : BAL_BINS - 1;
    atomicAdd(&h[c], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < BAL_BINS; k += 256) if (h[k]) atomicAdd(&hist[k], h[k]);
}
__global__ __launch_bounds__(64) void balance_scan_kernel(u32* __restrict__ hist) {   // descending exclusive scan of 1024 counters, one lane
    if (threadIdx.x | blockIdx.x) return;
    u32 acc = 0;
    for (int k = BAL_BINS - 1; k >= 0; --k) { const u32 v = hist[k]; hist[k] = acc; acc += v; }
}
__global__ __launch_bounds__(256) void balance_scatter_kernel(const u32* __restrict__ counts, u32* __restrict__ cursor, u32* __restrict__ order) {
    __shared__ u32 h[BAL_BINS];   // block-local ranks first: one device-scope atomic per (block, non-empty size), not per bucket
    for (int k = threadIdx.x; k < BAL_BINS; k += 256) h[k] = 0;
    __syncthreads();
    const u32 key = blockIdx.x * 256 + threadIdx.x;
    const u32 c = counts[key] < BAL_BINS ? counts[key] : BAL_BINS - 1;
    const u32 rank = atomicAdd(&h[c], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < BAL_BINS; k += 256) if (h[k]) h[k] = atomicAdd(&cursor[k], h[k]);
    __syncthreads();
    order[h[c] + rank] = key;
}
// FIRST: the buckets start empty; otherwise this launch adds a further chunk of the points to what the buckets hold
// G2 (Fq2 coordinates): the accumulator, the incoming point and the products' 64-bit columns are 300-340 VGPRs -- one wave per SIMD, and a lone
// wave issues at most every ~6.5 cycles even with independent instructions at hand (profiles/r05/ubench_lat.txt).  ZK_MSM_G2_WAVES=2 (variant
// builds: tools/build_variant.sh) caps the kernel at 256 registers for two waves per SIMD.
#undef MSM_ACC_BOUNDS
#define MSM_ACC_BOUNDS __launch_bounds__(64)
template <bool FIRST>
__global__ MSM_ACC_BOUNDS void msm_accumulate_kernel(const u32* __restrict__ conv, const u32* __restrict__ offsets,
                                                            const u32* __restrict__ counts, const u32* __restrict__ idx,
                                                            const u32* __restrict__ order, xyzz* __restrict__ buckets) {
    const u32 key = order[blockIdx.x * blockDim.x + threadIdx.x];  // window * 2^16 + digit, heaviest first
    xyzz acc = FIRST ? pt_inf() : buckets[key];
    const u32 n = counts[key], off = offsets[key];
    // the next point's coordinates (and the index after it) are requested before the current addition starts: the gather's
    // two dependent loads (index, then 2 x NR words at a random address) fly during ~2 500 instructions of arithmetic
    u32 i_next = n > 1 ? idx[off + 1] : 0;
    aff cur = n ? load_aff(conv, idx[off]) : aff{};
    for (u32 k = 0; k < n; ++k) {
        aff nxt = cur;
        if (k + 1 < n) nxt = load_aff(conv, i_next);
        if (k + 2 < n) i_next = idx[off + k + 2];
        acc = pt_madd(acc, cur);
        cur = nxt;
    }
    buckets[key] = acc;
}
// One level of the radix-R hierarchy (R = 2^RLOG) that computes sum_k k*B_k per window.  An item (S, A) stands
// for a block of m = R^level consecutive buckets: S = their sum, A = sum (local index) * bucket.
// R neighbouring blocks combine as S' = sum_j S_j, A' = sum_j A_j + m * sum_j j*S_j (running-sum
// trick for the last term, m = RLOG*level doublings).  Level 0 reads the buckets themselves (A = 0).  After
// 16/RLOG levels the one item left per window holds A = sum_k k*B_k.  The tail of a sum is the serial chain of
// these levels: 3R - 1 + RLOG*level point operations each -- 212 in all for R = 16 (first version), 144 for R = 4.
constexpr int RED_RLOG = 2;
// QUAD: four lanes per item (pt_add4 / pt_dbl4) -- for the upper levels, where a few hundred items leave the device idle and
// the serial chain of a lane is all that counts; the wide lower levels keep one lane per item (a quad does 1.7 x the work)
template <int RLOG, bool QUAD>
__global__ __launch_bounds__(64) void msm_reduce_level_kernel(const xyzz* __restrict__ S_in, const xyzz* __restrict__ A_in,
                                                              xyzz* __restrict__ S_out, xyzz* __restrict__ A_out,
                                                              u32 n_out, int level) {
    constexpr int R = 1 << RLOG;
    const u32 g = (blockIdx.x * blockDim.x + threadIdx.x) >> (QUAD ? 2 : 0);
    if (g >= n_out) return;
    auto add = [](const xyzz& a, const xyzz& b) { return QUAD ? pt_add4(a, b) : pt_add(a, b); };
    const xyzz* s = S_in + (u64)g * R;
    xyzz run = pt_inf(), acc = pt_inf();
    for (int j = R - 1; j >= 1; --j) { run = add(run, s[j]); acc = add(acc, run); }
    run = add(run, s[0]);
    if (level > 0) {
        for (int k = 0; k < RLOG * level; ++k) acc = QUAD ? pt_dbl4(acc) : pt_dbl(acc);
        const xyzz* a = A_in + (u64)g * R;
        for (int j = 0; j < R; ++j) acc = add(acc, a[j]);
    }
    if (!QUAD || (threadIdx.x & 3) == 0) { S_out[g] = run; A_out[g] = acc; }
}
__global__ __launch_bounds__(64) void msm_final_kernel(const xyzz* __restrict__ win, u32* __restrict__ out /* 2*CW_STD words + flag */, int weighted) {
    if (blockIdx.x || threadIdx.x >= 4) return;                            // one quad walks the windows
    xyzz acc = pt_inf();
    for (int w = N_WIN - 1; w >= 0; --w) {
        if (!weighted) for (int k = 0; k < C_BITS; ++k) acc = pt_dbl4(acc);   // weighted: the table already holds 2^(16 w) P
        acc = pt_add4(acc, win[w]);
    }
    if (threadIdx.x) return;
    if (pt_is_inf(acc)) { for (int i = 0; i < 2 * CW_STD; ++i) out[i] = 0; out[2 * CW_STD] = 1; return; }
    u32 x[CW_STD], y[CW_STD];
    pt_to_std(acc, x, y);
    for (int i = 0; i < CW_STD; ++i) { out[i] = x[i]; out[CW_STD + i] = y[i]; }
    out[2 * CW_STD] = 0;
}

// synthetic bases for benches/tests: P_i = [k_i]G, G = the curve's generator; k_i 64-bit, non-zero
__global__ __launch_bounds__(64) void g1_mul_generator_kernel(const u64* __restrict__ k, u64 n, u32* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 gx[CW_STD], gy[CW_STD];
    for (int j = 0; j < CW_STD; ++j) { gx[j] = GEN_X(j); gy[j] = GEN_Y(j); }
    aff g; g.x = cf_from_std(gx); g.y = cf_from_std(gy);
    const u64 e = k[i];
    xyzz acc = pt_inf();
    for (int b = 63; b >= 0; --b) {
        acc = pt_dbl(acc);
        if ((e >> b) & 1) acc = pt_madd(acc, g);
    }
    u32* o = out + i * (2 * CW_STD);
    if (pt_is_inf(acc)) { for (int j = 0; j < 2 * CW_STD; ++j) o[j] = 0; return; }
    u32 x[CW_STD], y[CW_STD];
    pt_to_std(acc, x, y);
    for (int j = 0; j < CW_STD; ++j) { o[j] = x[j]; o[CW_STD + j] = y[j]; }
}

void g1_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(g1_mul_generator_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_k, n, (u32*)d_bases);
    ZK_HIP(hipGetLastError());
}

#ifndef MSM_G2
// Fq elements in place: canonical integers (what bellman's key files and proof.json carry) <-> Montgomery R = 2^(32 NL)
__global__ __launch_bounds__(256) void fq_convert_kernel(u32* __restrict__ v, u64 n, int to_mont) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[NL];
    for (int k = 0; k < NL; ++k) w[k] = v[i * NL + k];
    if (to_mont) {
        fe x;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int bit = LB * k, wi = bit >> 5, s = bit & 31;
            u32 t = wi < NL ? w[wi] >> s : 0;
            if (s > 32 - LB && wi + 1 < NL) t |= w[wi + 1] << (32 - s);
            x.l[k] = t & LMASK;
        }
        fe c;
#pragma unroll
        for (int k = 0; k < NR; ++k) c.l[k] = RRP29(k);
        fe_to_std(fe_mul(x, c), w);
    } else {
        fe one = fe_zero(); one.l[0] = 1;
        const fe x = fe_canon(fe_mul(fe_from_std(w), one));
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int bit = 32 * j, k = bit / LB, s = bit % LB;
            u32 t = x.l[k] >> s;
            if (k + 1 < NR) t |= x.l[k + 1] << (LB - s);
            if (k + 2 < NR && 2 * LB - s < 32) t |= x.l[k + 2] << (2 * LB - s);
            w[j] = t;
        }
    }
    for (int k = 0; k < NL; ++k) v[i * NL + k] = w[k];
}
void fq_canon_to_mont_dev(void* d, uint64_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(fq_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u32*)d, n, 1);
    ZK_HIP(hipGetLastError());
}
void fq_mont_to_canon_dev(void* d, uint64_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(fq_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u32*)d, n, 0);
    ZK_HIP(hipGetLastError());
}
#endif

// ---- bucket reduction for window tables.  All 16 windows carry the same weight there, so buckets of equal digit merge
// ---- (M_d = sum_w B_{w,d}) and sum_d d M_d = sum_b 2^b T_b with T_b = the plain sum of the M_d whose digit has bit b
// ---- set.  When n is small this tail IS the sum (2^18 points: 2 ms of accumulation, 4.3 ms of tail on G1, 3.2 + 7.9 on G2 of
// ---- BLS12-381): every stage is a chain of point additions on lanes that have a SIMD to themselves, ~47 us each (G1, 14 limbs).
// ---- Round 5 cuts the chain from 96 point operations to 46: the merge on two lanes per digit (8 deep instead of 15), the T_b as trees
// ---- inside a workgroup (7 + 6 deep for 512 digits, then 6 for the 64 partial sums of a bit; were 8 + 4 x 7), and 2^b T_b by b
// ---- doublings in lane b followed by a 4-level tree (15 doublings + 4 additions; the Horner walk was 15 of each).
constexpr u32 BT_K = 8;                                                        // digits a lane sums before the workgroup's tree
constexpr u32 BT_CHUNKS = (N_BUCKET / 2) / (64 * BT_K);                        // partial sums per bit after the first level: 64
static_assert(BT_CHUNKS == 64 && C_BITS == 16, "the second level is one workgroup of 64 lanes per bit; the last one lane per bit");
// sum over the workgroup's 64 lanes, result in lane 0 (6 dependent additions); every lane must call
__device__ __forceinline__ xyzz wg_tree_sum(xyzz acc, xyzz* sh, u32 l, u32 width) {
    for (u32 s = width >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (l >= s && l < 2 * s) sh[l] = acc;
        __syncthreads();
        if (l < s) acc = pt_add(acc, sh[l + s]);
    }
    return acc;
}
__global__ __launch_bounds__(64) void msm_merge_windows_kernel(const xyzz* __restrict__ buckets, xyzz* __restrict__ merged) {
    __shared__ xyzz sh[32];
    constexpr u32 HW = N_WIN / 2;                                               // lane (digit, half): windows HW h .. HW h + HW - 1
    const u32 l = threadIdx.x, h = l >> 5, d = blockIdx.x * 32 + (l & 31);
    xyzz acc = buckets[(u32)(HW * h) * N_BUCKET + d];
    for (u32 w = 1; w < HW; ++w) acc = pt_add(acc, buckets[(u32)(HW * h + w) * N_BUCKET + d]);
    if (h) sh[l & 31] = acc;
    __syncthreads();
    if (!h) merged[d] = pt_add(acc, sh[l]);
}
static_assert(N_WIN % 2 == 0, "two halves of the windows");
// block (chunk, b): the 512 digits number 512 chunk .. among those with bit b set (k-th such digit: a 1 spliced into k at bit b)
__global__ __launch_bounds__(64) void msm_bit_tree_kernel(const xyzz* __restrict__ merged, xyzz* __restrict__ out /* [16][64] */) {
    __shared__ xyzz sh[64];
    const u32 l = threadIdx.x, b = blockIdx.y, k0 = (blockIdx.x * 64 + l) * BT_K;
    xyzz acc = pt_inf();
    for (u32 j = 0; j < BT_K; ++j) {
        const u32 k = k0 + j;
        const u32 d = ((k >> b) << (b + 1)) | (1u << b) | (k & ((1u << b) - 1));
        acc = pt_add(acc, merged[d]);
    }
    acc = wg_tree_sum(acc, sh, l, 64);
    if (l == 0) out[b * BT_CHUNKS + blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void msm_bit_sum_kernel(const xyzz* __restrict__ in /* [16][64] */, xyzz* __restrict__ T /* [16] */) {
    __shared__ xyzz sh[64];
    const u32 l = threadIdx.x, b = blockIdx.x;
    const xyzz acc = wg_tree_sum(in[b * BT_CHUNKS + l], sh, l, 64);
    if (l == 0) T[b] = acc;
}
__global__ __launch_bounds__(64) void msm_final_bits_kernel(const xyzz* __restrict__ T /* [16] */, u32* __restrict__ out) {
    __shared__ xyzz sh[64];
    const u32 l = threadIdx.x;
    xyzz acc = l < (u32)C_BITS ? T[l] : pt_inf();
    for (u32 k = 0; k < (u32)C_BITS - 1; ++k)
        if (k < l && l < (u32)C_BITS) acc = pt_dbl(acc);                      // lane b: 2^b T_b
    acc = wg_tree_sum(acc, sh, l, 16);
    if (l) return;
    if (pt_is_inf(acc)) { for (int i = 0; i < 2 * CW_STD; ++i) out[i] = 0; out[2 * CW_STD] = 1; return; }
    u32 x[CW_STD], y[CW_STD];
    pt_to_std(acc, x, y);
    for (int i = 0; i < CW_STD; ++i) { out[i] = x[i]; out[CW_STD + i] = y[i]; }
    out[2 * CW_STD] = 0;
}

// ---- window tables for bases that do not change between calls (a Groth16 proving key): entry [w * n + i] =
// ---- 2^(16 w) P_i in the internal affine layout.  The sum then needs no doublings at all: every (point, window)
// ---- pair adds the table entry of its window, and the 16 window results are simply added.
__global__ __launch_bounds__(64) void msm_table_kernel(u32* __restrict__ table, u64 n) {   // window 0 is in place already
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    aff a = load_aff(table, (u32)i);
    for (int w = 1; w < N_WIN; ++w) {
        xyzz p = pt_dbl_aff(a);
        for (int k = 1; k < C_BITS; ++k) p = pt_dbl(p);
        const cf izzz = cf_inv(p.ZZZ), t = cf_mul(p.ZZ, izzz), izz = cf_sqr(t);   // 1/ZZ = (ZZ/ZZZ)^2
        a.x = cf_mul(p.X, izz); a.y = cf_mul(p.Y, izzz);
        u32* o = table + ((u64)w * n + i) * PTW;
        cf_store_int(a.x, o); cf_store_int(a.y, o + CW_INT);
    }
}
size_t msm_fixed_table_bytes(uint64_t n) { return (size_t)n * N_WIN * PTW * 4; }
void msm_fixed_prepare_dev(const void* d_bases, uint64_t n, void* d_table, hipStream_t st) {
    ZK_REQUIRE(n >= 1 && n < (1ull << 24), "msm table: n out of range");
    hipLaunchKernelGGL(msm_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u32*)d_bases, n, (u32*)d_table);
    hipLaunchKernelGGL(msm_table_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (u32*)d_table, n);
    ZK_HIP(hipGetLastError());
}

// d_out: 2*CW_STD + 1 u32 words (x, y Montgomery, infinity flag).  d_table != nullptr: the sum runs over the n points
// [base_off, base_off + n) of a window table built for table_n points; d_bases is ignored
// d_preconv != nullptr: n points already in the internal layout (PTW words each); d_bases is ignored
// one non-blocking stream per device and host thread for the sorts that run beside an accumulation (msm_core)
static hipStream_t msm_side_stream() {
    thread_local hipStream_t ss[64] = {};
    int dev = 0; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    if (!ss[dev]) ZK_HIP(hipStreamCreateWithFlags(&ss[dev], hipStreamNonBlocking));
    return ss[dev];
}
static void msm_core(const void* d_bases, const void* d_table, uint64_t table_n, uint64_t base_off, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st,
                     const void* d_preconv = nullptr) {
    ZK_REQUIRE(n >= 1 && n < (1ull << 28), "msm: n out of range");
    ZK_REQUIRE(!d_table || (n < (1ull << 24) && base_off + n <= table_n && table_n < (1ull << 24)), "msm table: range out of bounds");
    const size_t n_keys = (size_t)N_WIN * N_BUCKET;
    DevBuf counts, offsets, cursors, tops, idx, buckets, S0, A0, S1, A1, conv;
    if (!d_table && !d_preconv) {
        conv.reserve((size_t)n * PTW * 4);
        hipLaunchKernelGGL(msm_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u32*)d_bases, n, (u32*)conv.p);
        ZK_HIP(hipGetLastError());
    }
    const u32* points = d_table ? (const u32*)d_table : d_preconv ? (const u32*)d_preconv : (const u32*)conv.p;
    tops.reserve(1024 * 4);
    buckets.reserve(n_keys * sizeof(xyzz));
    S0.reserve((n_keys >> RED_RLOG) * sizeof(xyzz)); A0.reserve((n_keys >> RED_RLOG) * sizeof(xyzz));   // >= N_BUCKET items: the table path's merged / partial arrays fit
    S1.reserve(std::max<size_t>((size_t)C_BITS * 512, n_keys >> (2 * RED_RLOG)) * sizeof(xyzz)); A1.reserve((n_keys >> (2 * RED_RLOG)) * sizeof(xyzz));   // S1 also serves the bit-partial levels (16 x 512 items)
    // The points go through sort -> accumulate in chunks: the sort of chunk c + 1 (memory-bound: two partition passes over its
    // pairs, 1.6 ms per 2^23 points) runs on a side stream beside the accumulation of chunk c (integer-ALU bound, 5.7 ms), and every
    // later chunk adds to the buckets the earlier ones left.  One chunk for small sums, for the table path and for the atomic sort.
    static const int chunks_env = getenv("ZK_MSM_CHUNKS") ? atoi(getenv("ZK_MSM_CHUNKS")) : 4;
    // (measured, BN254 G1 behind the endomorphism split: 2^22 points 8.80 -> 8.25 ms with 4 chunks, 8.49 with 2, 8.54 with 8; 2^20 points
    // 2.88 -> 3.38 ms: short sums are the latency of their launch chain, which chunking lengthens)
    // Round 4: sums of 2^24 and more pairs go through the same chunks -- of 2^22 pairs each, so that every chunk takes the LDS-histogram
    // sort (its point indices are chunk-relative, 24 bits) instead of ONE sort with device-scope atomics over all pairs:
    // 2^24 points (2^25 pairs) 48.1 -> see profiles/r04/msm_large.txt.
    // Round 6: chunks that GROW (2^18, 2^19, ... pairs, so that the first sort, which nothing hides, is short) were measured and are slower:
    // BN254 2^22 points 8.17-8.28 ms against 7.97-8.08 with four equal chunks, BLS12-381 16.75-16.86 against 16.59-16.68 -- every
    // accumulation launch walks all 2^19 buckets, and a chunk that leaves 0-4 points per bucket pays the walk for nothing.
    int n_chunks = 1;
    if (!d_table && chunks_env > 1) {
        if (n >= (1ull << 23) && n < (1ull << 24)) n_chunks = std::min(chunks_env, 8);
        else if (n >= (1ull << 24)) n_chunks = (int)((n + (1ull << 22) - 1) >> 22);
    }
    const u64 chunk_n = ((n + n_chunks - 1) / n_chunks + 255) / 256 * 256;
    const bool lds_sort = chunk_n < (1ull << 24);          // the partition sort's indices are relative to the chunk
    struct Chunk { DevBuf counts, offsets, order, idx; };
    std::vector<Chunk> CH(n_chunks);
    DevBuf hist, hist_scanned, coarse, bal;
    bal.reserve(BAL_BINS * 4);
    hipStream_t ss = st;                                   // the stream the sorts run on
    std::vector<hipEvent_t> ev_sorted;
    hipEvent_t ev_ready = nullptr;
    if (n_chunks > 1) {
        ss = msm_side_stream();
        on_stream(st); on_side_stream(ss);                 // the pool orders this thread's frees behind both streams while the side stream is in use
        for (int c = 0; c < n_chunks; ++c) { hipEvent_t e; ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_sorted.push_back(e); }
        ZK_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
    }
    struct EvGuard { std::vector<hipEvent_t>& v; hipEvent_t& r; ~EvGuard() { for (hipEvent_t e : v) (void)hipEventDestroy(e); if (r) (void)hipEventDestroy(r); } } ev_guard{ev_sorted, ev_ready};
    for (int c = 0; c < n_chunks; ++c) {                   // every buffer exists before the side stream starts: it waits for ONE event of `st`
        const u64 c0 = (u64)c * chunk_n, nc = c0 < n ? std::min<u64>(chunk_n, n - c0) : 0;
        CH[c].counts.reserve(n_keys * 4); CH[c].offsets.reserve(n_keys * 4); CH[c].order.reserve(n_keys * 4);
        CH[c].idx.reserve(std::max<u64>(1, nc) * N_WIN * 4);
    }
    if (lds_sort) {
        const u32 nb_max = (u32)((chunk_n + SORT_PTS - 1) / SORT_PTS);
        const size_t n_hist_max = ((size_t)N_COARSE * nb_max + 1023) / 1024 * 1024;
        hist.reserve(n_hist_max * 4); hist_scanned.reserve(n_hist_max * 4); coarse.reserve(chunk_n * N_WIN * 4);
        tops.reserve(n_hist_max / 1024 * 4 + 4);
    } else cursors.reserve(n_keys * 4);
    if (n_chunks > 1) { ZK_HIP(hipEventRecord(ev_ready, st)); ZK_HIP(hipStreamWaitEvent(ss, ev_ready, 0)); }   // (the points: converted / split on `st`)
    for (int c = 0; c < n_chunks; ++c) {
        const u64 c0 = (u64)c * chunk_n, nc = c0 < n ? std::min<u64>(chunk_n, n - c0) : 0;
        const u32* sc = (const u32*)d_scalars + c0 * MSM_SC_WORDS;
        u32 *counts_p = (u32*)CH[c].counts.p, *offsets_p = (u32*)CH[c].offsets.p, *idx_p = (u32*)CH[c].idx.p, *order_p = (u32*)CH[c].order.p;
        const u64 total = nc * N_WIN;
        if (nc == 0) { ZK_HIP(hipMemsetAsync(counts_p, 0, n_keys * 4, ss)); ZK_HIP(hipMemsetAsync(offsets_p, 0, n_keys * 4, ss)); }
        else if (lds_sort) {  // LDS-histogram partition (no device-scope atomics)
            const u32 n_blocks = (u32)((nc + SORT_PTS - 1) / SORT_PTS);
            const size_t n_hist = ((size_t)N_COARSE * n_blocks + 1023) / 1024 * 1024;   // scan granularity
            ZK_HIP(hipMemsetAsync(hist.p, 0, n_hist * 4, ss));
            hipLaunchKernelGGL(sort_hist_kernel, dim3(n_blocks), dim3(256), 0, ss, sc, nc, n_blocks, (u32*)hist.p);
            const unsigned nb = (unsigned)(n_hist / 1024);
            hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(256), 0, ss, (const u32*)hist.p, (u32*)hist_scanned.p, (u32*)tops.p);
            hipLaunchKernelGGL(scan_tops_kernel, dim3(1), dim3(64), 0, ss, (u32*)tops.p, nb);
            hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, ss, (u32*)hist_scanned.p, (const u32*)tops.p);
            ZK_HIP(hipGetLastError());
            hipLaunchKernelGGL(sort_coarse_kernel, dim3(n_blocks), dim3(256), 0, ss, sc, nc, n_blocks, (const u32*)hist_scanned.p, (u32*)coarse.p);
            // the last coarse bin ends at the number of non-zero pairs: last scanned entry + last count, kept on the device
            hipLaunchKernelGGL(sort_total_kernel, dim3(1), dim3(64), 0, ss, (const u32*)hist.p, (const u32*)hist_scanned.p, (u32)(n_hist - 1), (u32*)tops.p);
            hipLaunchKernelGGL(sort_fine_kernel2, dim3(N_COARSE), dim3(256), 0, ss, (const u32*)coarse.p, (const u32*)hist_scanned.p, n_blocks, (const u32*)tops.p,
                               counts_p, offsets_p, idx_p, d_table ? (u32)table_n : 0u, d_table ? (u32)base_off : 0u);
            ZK_HIP(hipGetLastError());
        } else {
            ZK_HIP(hipMemsetAsync(counts_p, 0, n_keys * 4, ss));
            ZK_HIP(hipMemsetAsync(cursors.p, 0, n_keys * 4, ss));
            const unsigned gb = (unsigned)((total + 255) / 256);
            hipLaunchKernelGGL(msm_count_kernel, dim3(gb), dim3(256), 0, ss, sc, nc, counts_p);
            ZK_HIP(hipGetLastError());
            const unsigned nb = (unsigned)(n_keys / 1024);
            hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(256), 0, ss, (const u32*)counts_p, offsets_p, (u32*)tops.p);
            hipLaunchKernelGGL(scan_tops_kernel, dim3(1), dim3(64), 0, ss, (u32*)tops.p, nb);
            hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, ss, offsets_p, (const u32*)tops.p);
            ZK_HIP(hipGetLastError());
            hipLaunchKernelGGL(msm_scatter_kernel, dim3(gb), dim3(256), 0, ss, sc, nc, (const u32*)offsets_p, (u32*)cursors.p, idx_p);
            ZK_HIP(hipGetLastError());
        }
        ZK_HIP(hipMemsetAsync(bal.p, 0, BAL_BINS * 4, ss));
        hipLaunchKernelGGL(balance_hist_kernel, dim3((unsigned)(n_keys / 256)), dim3(256), 0, ss, (const u32*)counts_p, (u32*)bal.p);
        hipLaunchKernelGGL(balance_scan_kernel, dim3(1), dim3(64), 0, ss, (u32*)bal.p);
        hipLaunchKernelGGL(balance_scatter_kernel, dim3((unsigned)(n_keys / 256)), dim3(256), 0, ss, (const u32*)counts_p, (u32*)bal.p, order_p);
        ZK_HIP(hipGetLastError());
        if (n_chunks > 1) { ZK_HIP(hipEventRecord(ev_sorted[c], ss)); ZK_HIP(hipStreamWaitEvent(st, ev_sorted[c], 0)); }
        const u32* pts_c = points + (d_table ? 0 : c0 * PTW);             // (chunk-relative indices; the table path has one chunk)
        if (c == 0) hipLaunchKernelGGL((msm_accumulate_kernel<true>), dim3((unsigned)(n_keys / 64)), dim3(64), 0, st, pts_c,
                                       (const u32*)offsets_p, (const u32*)counts_p, (const u32*)idx_p, (const u32*)order_p, (xyzz*)buckets.p);
        else hipLaunchKernelGGL((msm_accumulate_kernel<false>), dim3((unsigned)(n_keys / 64)), dim3(64), 0, st, pts_c,
                                (const u32*)offsets_p, (const u32*)counts_p, (const u32*)idx_p, (const u32*)order_p, (xyzz*)buckets.p);
        ZK_HIP(hipGetLastError());
    }
    // `st` has waited for the side stream's last event: everything it did is ordered before what follows on `st`, so the pool may forget
    // it (frees are stamped on `st` alone again, and a process that only ever used the null stream is back on the event-free fast path)
    if (n_chunks > 1) forget_stream(ss);
    if (d_table) {   // equal window weights: merge, 16 trees over the digits with a bit set, 2^b T_b in lane b and a last tree
        xyzz* merged = (xyzz*)S0.p;                       // N_BUCKET items fit: S0 holds n_keys / 16 = N_BUCKET
        xyzz* pa = (xyzz*)A0.p; xyzz* pb = (xyzz*)S1.p;   // 16 x 64 partial sums, then the 16 T_b
        hipLaunchKernelGGL(msm_merge_windows_kernel, dim3(N_BUCKET / 32), dim3(64), 0, st, (const xyzz*)buckets.p, merged);
        hipLaunchKernelGGL(msm_bit_tree_kernel, dim3(BT_CHUNKS, C_BITS), dim3(64), 0, st, (const xyzz*)merged, pa);
        hipLaunchKernelGGL(msm_bit_sum_kernel, dim3(C_BITS), dim3(64), 0, st, (const xyzz*)pa, pb);
        hipLaunchKernelGGL(msm_final_bits_kernel, dim3(1), dim3(64), 0, st, (const xyzz*)pb, (u32*)d_out);
        ZK_HIP(hipGetLastError());
        return;
    }
    // four lanes per item from the level with this many items up (measured: G1 gains from 2^15 items on; over Fq2 the selects and
    // broadcasts of a stage cost what the shorter chain saves, only the final walk keeps its quad)
#ifdef MSM_G2
    constexpr u32 quad_default = 0;
#else
    constexpr u32 quad_default = 32768;
#endif
    static const u32 quad_below = getenv("ZK_MSM_QUAD_BELOW") ? (u32)atoi(getenv("ZK_MSM_QUAD_BELOW")) : quad_default;
    // reduction hierarchy: ping-pong (S, A) arrays of n_keys / R and n_keys / R^2 items
    const xyzz* s_in = (const xyzz*)buckets.p; const xyzz* a_in = nullptr;
    u32 n_out = (u32)(n_keys >> RED_RLOG);
    for (int level = 0; level < C_BITS / RED_RLOG; ++level, n_out >>= RED_RLOG) {
        xyzz* s_out = (xyzz*)(level & 1 ? S1.p : S0.p); xyzz* a_out = (xyzz*)(level & 1 ? A1.p : A0.p);
        if (n_out <= quad_below)
            hipLaunchKernelGGL((msm_reduce_level_kernel<RED_RLOG, true>), dim3((4 * n_out + 63) / 64), dim3(64), 0, st, s_in, a_in, s_out, a_out, n_out, level);
        else
            hipLaunchKernelGGL((msm_reduce_level_kernel<RED_RLOG, false>), dim3((n_out + 63) / 64), dim3(64), 0, st, s_in, a_in, s_out, a_out, n_out, level);
        s_in = s_out; a_in = a_out;
    }
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(msm_final_kernel, dim3(1), dim3(64), 0, st, a_in, (u32*)d_out, d_table ? 1 : 0);
    ZK_HIP(hipGetLastError());
}   // the pooled scratch above is released here, stream-ordered: the sum is asynchronous on `st` like every other _dev entry point

void msm_preconv_dev(const void* d_points, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) { msm_core(nullptr, nullptr, 0, 0, d_scalars, n, d_out, st, d_points); }
#ifdef MSM_GLV
// ---- the curve's endomorphism phi(x, y) = (beta x, y) = [lambda](x, y): k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^128, so a sum
// over n points with 254-bit scalars becomes a sum over 2n points with 128-bit scalars: the same number of bucket additions, half
// the windows -- half the bucket hierarchy and half of the doublings of the final Horner walk, the two serial tails of a sum.
// (k1, k2) = k - c1 (a1, b1) - c2 (a2, b2) with c1 = floor(k g1 / 2^256), c2 = floor(k g2 / 2^256), g1 = floor(2^256 b2 / r),
// g2 = floor(-2^256 b1 / r) for the short basis (a1, b1), (a2, b2) of {(x, y): x + y lambda = 0 mod r}; any integers c1, c2 give
// a correct split, these keep both halves below 2^128 (checked over the edge scalars and 2 * 10^5 random ones when the constants
// were derived, tools/glv_constants.py).  One kernel splits the scalar, converts the base and writes (+-P, |k1|), (+-phi(P), |k2|).
template <int NA, int NB, int NO>
__device__ __forceinline__ void glv_mul(const u32 (&a)[NA], const u32 (&b)[NB], u32 (&out)[NO], int from) {   // words [from, from + NO) of a * b
    u64 acc = 0;
    u32 carry_hi = 0;
    for (int k = 0; k < from + NO; ++k) {       // column k; (acc, carry_hi) is a 96-bit running sum
        for (int i = 0; i < NA; ++i) {
            const int j = k - i;
            if (j < 0 || j >= NB) continue;
            const u64 p = (u64)a[i] * b[j];
            acc += p;
            carry_hi += acc < p;
        }
        if (k >= from) out[k - from] = (u32)acc;
        acc = (acc >> 32) | ((u64)carry_hi << 32);
        carry_hi = 0;
    }
}
__global__ __launch_bounds__(256) void glv_split_kernel(const u32* __restrict__ bases, const u32* __restrict__ scalars, u64 n,
                                                        u32* __restrict__ conv2 /* 2n points */, u32* __restrict__ sc2 /* 2n x 4 words */) {
    __shared__ cf beta_s;                                              // beta in the internal form, converted once per block
    if (threadIdx.x == 0) { const u32 BETA[NL] = {GLV_BETA_STD}; beta_s = cf_from_std(BETA); }
    __syncthreads();
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = scalars[i * 8 + j];
    bool n1 = false, n2 = false;
#ifdef GLV_LAMBDA
    // lambda^2 + lambda + 1 = 0 (mod r) with lambda < 2^128: k = k1 + k2 lambda by plain division, both halves non-negative.  A scalar
    // is brought below r first (k2 <= lambda + 1 needs it); the quotient from the reciprocal g = floor(2^256 / lambda) is at most
    // one short, made up by one conditional step
    const u32 LAMBDA[4] = {GLV_LAMBDA}, GG[5] = {GLV_G}, RMOD[8] = {GLV_R};
    for (int rep = 0; rep < 2; ++rep) {
        u32 t[8]; u64 br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u64 d = (u64)k[j] - RMOD[j] - br; t[j] = (u32)d; br = (d >> 32) & 1; }
        if (!br) { for (int j = 0; j < 8; ++j) k[j] = t[j]; }
    }
    u32 c[5], t8[8];
    glv_mul<8, 5, 5>(k, GG, c, 8);
    glv_mul<5, 4, 8>(c, LAMBDA, t8, 0);
    u64 br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const u64 d = (u64)k[j] - t8[j] - br; k1[j] = (u32)d; br = (d >> 32) & 1; }
#pragma unroll
    for (int j = 0; j < 8; ++j) k2[j] = j < 5 ? c[j] : 0;
    {   // k1 >= lambda: one more lambda goes to k2
        u32 t[8]; u64 b2 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u64 d = (u64)k1[j] - (j < 4 ? LAMBDA[j] : 0u) - b2; t[j] = (u32)d; b2 = (d >> 32) & 1; }
        if (!b2) {
            for (int j = 0; j < 8; ++j) k1[j] = t[j];
            u64 cy = 1;
            for (int j = 0; j < 8; ++j) { cy += k2[j]; k2[j] = (u32)cy; cy >>= 32; }
        }
    }
#else
    const u32 G1[3] = {GLV_G1}, G2[5] = {GLV_G2}, A1[2] = {GLV_A1}, A2[4] = {GLV_A2}, NB1[4] = {GLV_NB1}, B2[2] = {GLV_B2};
    u32 c1[3], c2[5];
    glv_mul<8, 3, 3>(k, G1, c1, 8);
    glv_mul<8, 5, 5>(k, G2, c2, 8);
    u32 t1[8], t2[8];
    glv_mul<3, 2, 8>(c1, A1, t1, 0); glv_mul<5, 4, 8>(c2, A2, t2, 0);          // k1 = k - c1 a1 - c2 a2  (mod 2^256, two's complement)
    u64 br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const u64 d = (u64)k[j] - t1[j] - t2[j] - br; k1[j] = (u32)d; br = (0 - (d >> 32)) & 3; }
    glv_mul<3, 4, 8>(c1, NB1, t1, 0); glv_mul<5, 2, 8>(c2, B2, t2, 0);         // k2 = c1 |b1| - c2 b2
    br = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const u64 d = (u64)t1[j] - t2[j] - br; k2[j] = (u32)d; br = (0 - (d >> 32)) & 1; }
    n1 = k1[7] >> 31; n2 = k2[7] >> 31;
    if (n1) { u64 c = 1; for (int j = 0; j < 8; ++j) { c += (u32)~k1[j]; k1[j] = (u32)c; c >>= 32; } }
    if (n2) { u64 c = 1; for (int j = 0; j < 8; ++j) { c += (u32)~k2[j]; k2[j] = (u32)c; c >>= 32; } }
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc2[(2 * i) * 4 + j] = k1[j]; sc2[(2 * i + 1) * 4 + j] = k2[j]; }
    u32 w[2 * CW_STD];
    const uint4* p = (const uint4*)(bases + i * (2 * CW_STD));
#pragma unroll
    for (int q = 0; q < 2 * CW_STD / 4; ++q) { const uint4 v = p[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    const cf x = cf_from_std(w), y = cf_from_std(w + CW_STD), ny = cf_sub<2>(cf_zero(), y);
    const cf bx = cf_mul(x, beta_s);
    u32* o = conv2 + (2 * i) * PTW;
    cf_store_int(x, o); cf_store_int(n1 ? ny : y, o + CW_INT);
    cf_store_int(bx, o + PTW); cf_store_int(n2 ? ny : y, o + PTW + CW_INT);
}
void msm_g1_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    if (n < 4096 || n >= (1ull << 27)) { msm_core(d_bases, nullptr, 0, 0, d_scalars, n, d_out, st); return; }
    DevBuf conv2, sc2;
    conv2.reserve((size_t)2 * n * PTW * 4); sc2.reserve((size_t)2 * n * 16);
    // the split of all points in front of the sum (made chunk by chunk beside the accumulation it gains nothing: the split is integer work
    // like the accumulation -- round 5, tools/experiments/rejected_switches_r02_r05.patch, profiles/r05/msm_ab_raw.txt)
    hipLaunchKernelGGL(glv_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u32*)d_bases, (const u32*)d_scalars, n, (u32*)conv2.p, (u32*)sc2.p);
    ZK_HIP(hipGetLastError());
    MSM_GLV::msm_preconv_dev(conv2.p, sc2.p, 2 * n, d_out, st);     // asynchronous: conv2 / sc2 go back to the pool with the sum still in flight; pool_free's events order their reuse
}
#else
void msm_g1_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) { msm_core(d_bases, nullptr, 0, 0, d_scalars, n, d_out, st); }
#endif
void msm_fixed_dev(const void* d_table, uint64_t table_n, uint64_t base_off, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    ZK_REQUIRE(d_table, "msm table: null table");
    msm_core(nullptr, d_table, table_n, base_off, d_scalars, n, d_out, st);
}
#undef MSM_N_WIN
#undef MSM_SC_WORDS
