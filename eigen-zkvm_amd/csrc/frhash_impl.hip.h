// Field-generic body of the scalar-field hashing (see frhash.hip): included once per field inside that field's
// namespace, which provides the fe29 constants (NL, NR, Q29, ...), RRP29 (R'^2 mod r), FH_NRP (partial rounds of
// t = 2..17), FH_OUT_IDX (the state word Poseidon::hash returns: 0 for BN128, 1 for BLS12-381), FH_NAME and
// FH_FN(name) (host entry points).  No include guard on purpose.
#ifndef ZK_FRHASH_MUL_ATTR
#define ZK_FRHASH_MUL_ATTR __forceinline__
#endif
#define FQ_MUL_ATTR ZK_FRHASH_MUL_ATTR   // inlined into the (not unrolled) loops over t: a dozen call sites per kernel

namespace {
#include "fe29_impl.hip.h"

// canonical 256-bit integer (8 words, may exceed r) -> internal Montgomery form: x * R'^2 / R' = x R'
__device__ __forceinline__ fe fe_from_int(const u32 (&w)[NL]) {
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
    for (int i = 0; i < NR; ++i) c.l[i] = RRP29(i);
    return fe_mul(x, c);
}
__device__ __forceinline__ fe fe_renorm(const fe& a) { return fe_mul(a, fe_one()); }   // value < 168r -> < 2r, same residue

// ---- parameter tables: per t (index t-2) offsets into one array of fe --------------------------------
struct Params { u32 t, n_rp; const fe* c; const fe* m; const fe* p; const fe* s;
                const u32* coef; const u32* rebw; u32 b1;
                const void* mf_m; const void* mf_p; const u64* mf_k;           // matrix-pipe form of the dense layers (fr_mfma.hip.h): fragments of M and P, addends [8 layers][t][8]
                const void* mf_s; const void* mf_i; };                          // ... and of the sparse rounds: n_rp blocks of mf_sparse_round_bytes(t), the unit's fragment   // cooperative form of the sparse rounds (coop_tables_kernel): [round][limb][lane], [phase][word - 1][limb][lane], first round of phase 1
__device__ Params g_prm[16];

__device__ __forceinline__ void pow5(fe& x) { const fe x2 = fe_sqr(x), x4 = fe_sqr(x2); x = fe_mul(x4, x); }  // poseidon_bn128_opt.rs:88-94

// Sums of products with deferred reduction (fe29_impl.hip.h fe_wide): the tables are canonical (< r), so a group of g
// products with operands < B r needs g B <= FH_AB_LIMIT (168 for BN254's r, 68 for BLS12-381's).
constexpr u32 DOT_DENSE = FE_WIDE_MAX;                                       // dense products: state < 3r (6 x 3 = 18)
constexpr u32 CO_RENORM = 5;                                                 // fused product-accumulates between two renormalisations (register and cooperative forms)
constexpr u32 PR_RENORM = 8;                                                 // sparse rounds: running columns renormalised every 8 rounds,
constexpr u32 DOT_SPARSE = FH_AB_LIMIT / 18 < FE_WIDE_MAX ? FH_AB_LIMIT / 18 : FE_WIDE_MAX;   // < 2r + 8 * 2r = 18r in between
// sum_{j < n} a[j * stride] * x[j] in groups of `grp`; n > grp: partial results (< 2r each) are added and renormalised
template <class GetX>
__device__ __forceinline__ fe dot_products(const fe* __restrict__ a, u32 stride, GetX getx, u32 n, u32 grp) {
    fe_wide w; fe_wide_zero(w);
    fe acc = fe_zero();
    u32 cnt = 0;
    for (u32 j = 0; j < n; ++j) {
        fe_wide_mac(w, a[(size_t)j * stride], getx(j));
        if (++cnt == grp || j + 1 == n) { acc = fe_add(acc, fe_wide_reduce(w)); cnt = 0; fe_wide_zero(w); }
    }
    return n > grp ? fe_renorm(acc) : acc;                                   // <= 9 groups: < 18r -> < 2r
}
// st <- (sum_j MAT[j][i] st[j])_i, st[j] < 3r in, < 2r out
__device__ void matmul(const fe* __restrict__ mat, fe* st, fe* tmp, u32 t) {
    for (u32 i = 0; i < t; ++i) tmp[i] = dot_products(mat + i, t, [&](u32 j) { return st[j]; }, t, DOT_DENSE);
    for (u32 i = 0; i < t; ++i) st[i] = tmp[i];
}

// poseidon_bn128_opt.rs:98-224 hash_inner on st[0..t) (st[0] = init state, st[1..] = inputs), all < 2r
__device__ __noinline__ void poseidon_fr(fe* st, fe* tmp, u32 t) {   // one generic copy: specialised for t = 17 it takes 256 VGPRs
    const Params P = g_prm[t - 2];
    for (u32 i = 0; i < t; ++i) st[i] = fe_add(st[i], P.c[i]);
    for (u32 r = 0; r < 3; ++r) {
        for (u32 i = 0; i < t; ++i) { pow5(st[i]); st[i] = fe_add(st[i], P.c[(r + 1) * t + i]); }
        matmul(P.m, st, tmp, t);
    }
    for (u32 i = 0; i < t; ++i) { pow5(st[i]); st[i] = fe_add(st[i], P.c[4 * t + i]); }
    matmul(P.p, st, tmp, t);
    for (u32 r = 0; r < P.n_rp; ++r) {
        pow5(st[0]);
        st[0] = fe_add(st[0], P.c[5 * t + r]);
        const fe* __restrict__ S = P.s + (size_t)(2 * t - 1) * r;
        const fe s0 = dot_products(S, 1, [&](u32 j) { return st[j]; }, t, DOT_SPARSE);   // st[j] < 18r
        for (u32 k = 1; k < t; ++k) st[k] = fe_add(st[k], fe_mul(S[t + k - 1], st[0])); // grows by < 2r per round
        st[0] = s0;
        if (r % PR_RENORM == PR_RENORM - 1) for (u32 k = 1; k < t; ++k) st[k] = fe_renorm(st[k]);   // < 2r + 8 * 2r in between
    }
    for (u32 k = 1; k < t; ++k) st[k] = fe_renorm(st[k]);
    for (u32 r = 0; r < 3; ++r) {
        for (u32 i = 0; i < t; ++i) { pow5(st[i]); st[i] = fe_add(st[i], P.c[5 * t + P.n_rp + r * t + i]); }
        matmul(P.m, st, tmp, t);
    }
    for (u32 i = 0; i < t; ++i) pow5(st[i]);
    matmul(P.m, st, tmp, t);
}

__device__ __forceinline__ fe load_raw(const u64* __restrict__ p) {   // 4 raw limbs -> internal
    u32 w[NL];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const u64 v = p[k]; w[2 * k] = (u32)v; w[2 * k + 1] = (u32)(v >> 32); }
    return fe_from_std(w);
}
__device__ __forceinline__ void store_raw(const fe& a, u64* __restrict__ p) {
    u32 w[NL];
    fe_to_std(a, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = ((u64)w[2 * k + 1] << 32) | w[2 * k];
}
// digest.rs:162-175 to_bn128 / linearhash_bn128.rs:70-91 to_bn128_mont: e0 + e1 2^64 + ... as a field element
__device__ __forceinline__ fe words_to_fe(const u64* __restrict__ e, u32 n) {
    u32 w[NL];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const u64 v = (u32)k < n ? e[k] : 0; w[2 * k] = (u32)v; w[2 * k + 1] = (u32)(v >> 32); }
    return fe_from_int(w);
}

// table conversion: canonical 32-byte integers -> internal form, in place layout change (8 -> NR words)
__global__ void bn128_convert_kernel(const u32* __restrict__ canon, u64 n, fe* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[NL];
    for (int k = 0; k < NL; ++k) w[k] = canon[i * NL + k];
    out[i] = fe_canon(fe_from_int(w));   // < r: the group bounds of dot_products count on it
}
// Poseidon::hash_ex on a batch: inp [n][n_in][4] raw, init [4] raw (shared), out [n][n_out][4] raw
__global__ __launch_bounds__(64) void bn128_poseidon_kernel(const u64* __restrict__ inp, u64 n, u32 n_in, const u64* __restrict__ init,
                                                            u32 n_out, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe st[17], tmp[17];
    st[0] = load_raw(init);
    for (u32 k = 0; k < n_in; ++k) st[k + 1] = load_raw(inp + (i * n_in + k) * 4);
    poseidon_fr(st, tmp, n_in + 1);
    for (u32 k = 0; k < n_out; ++k) store_raw(st[k], out + (i * n_out + k) * 4);
}
// LinearHashBN128::hash_element_array per row (linearhash_bn128.rs:105-131)
__global__ __launch_bounds__(64) void bn128_leaf_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= height) return;
    const u64* __restrict__ v = rows + i * width;
    if (width <= 4) { store_raw(words_to_fe(v, width), digests + 4 * i); return; }
    fe st[17], tmp[17];
    const u32 nb = (width - 1) / 3 + 1;
    fe digest = fe_zero();
    for (u32 b = 0; b < nb; b += 16) {
        const u32 sz = nb - b < 16 ? nb - b : 16;
        st[0] = digest;
        for (u32 k = 0; k < sz; ++k) {
            const u32 at = 3 * (b + k), len = width - at < 3 ? width - at : 3;
            st[k + 1] = words_to_fe(v + at, len);
        }
        poseidon_fr(st, tmp, sz + 1);
        digest = st[FH_OUT_IDX];                          // Poseidon::hash
    }
    store_raw(digest, digests + 4 * i);
}
// ---- the same permutation with the state in registers, for the t <= 9 of a row hash of up to 24 columns (the leaves
// ---- are four fifths of a tree's permutations).  Static indexing only: the loops over the t words are either fully
// ---- unrolled (dense column sums, sparse rounds) or rolled with the array rotated by one word per trip (S-box layer,
// ---- columns of the matrix product), so one copy of every product serves all words.  The eight full rounds share one
// ---- loop body; the sparse rounds sit between its fourth and fifth trip.  Same operation order and renormalisation
// ---- schedule as poseidon_fr, hence the same bounds.
template <int B, int E, class F>
__device__ __forceinline__ void fh_static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); fh_static_for<B + 1, E>(f); }
}
#ifndef ZK_FR_MFMA_FROM
#define ZK_FR_MFMA_FROM 3    // dense layers of t >= this on the matrix pipe (A/B knob; 18 = never)
#endif
#include "fr_mfma.hip.h"
template <int T>
__device__ __forceinline__ void reg_rotate_in(fe (&a)[T], const fe& last) {
    fh_static_for<0, T - 1>([&](auto I) { a[decltype(I)::value] = a[decltype(I)::value + 1]; });
    a[T - 1] = last;
}
template <int T, bool RENORM>
__device__ __forceinline__ fe reg_dot(const fe* __restrict__ a, u32 stride, const fe (&x)[T], u32 grp) {
    fe_wide w; fe_wide_zero(w);
    fe acc = fe_zero();
    u32 cnt = 0;
    fh_static_for<0, T>([&](auto J) {
        constexpr int j = decltype(J)::value;
        fe_wide_mac(w, a[(size_t)j * stride], x[j]);
        if (++cnt == grp || j + 1 == T) { acc = fe_add(acc, fe_wide_reduce(w)); cnt = 0; fe_wide_zero(w); }
    });
    return RENORM && (u32)T > grp ? fe_renorm(acc) : acc;                    // up to three groups: < 6r without
}
template <int T>
__device__ __forceinline__ void reg_renorm_tail(fe (&st)[T]) {               // st[1..T) < 2r, one rolled product
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int k = 1; k < T; ++k) {
        const fe v = fe_renorm(st[1]);
        fh_static_for<1, T - 1>([&](auto I) { st[decltype(I)::value] = st[decltype(I)::value + 1]; });
        st[T - 1] = v;
    }
}
// (The rolled loops carry interleave(disable) besides unroll(disable): with an even trip count the loop vectoriser otherwise
// interleaves two trips, doubling the live state -- T = 8 took 512 registers, T = 10 and up spilled to scratch.)
// Bounds (units of r).  Dense rounds: S-box output < 2, + constant < 3, a column of <= 3 groups of 6 products < 6 (no
// renormalisation: the next S-box squares it, 36 <= FH_AB_LIMIT).  Sparse rounds: the words 1..T-1 enter below 2 (one pass of
// products by one), each round adds S'[k] y to them inside the product's own reduction (fe_mul_acc: + < 1.05 a round), and after
// every CO_RENORM rounds they come back below 2 -- in a loop of its own: written as `if (r % 8 == 7)` inside the column loop
// the renormalising product was if-converted and ran every time (a quarter of the kernel).  The row product takes groups of
// 6 x 7.3 <= FH_AB_LIMIT.  The column update is unrolled: rolled, rotating the state cost 122 moves per 212-instruction product.
constexpr u32 DOT_SPARSE_REG = FH_AB_LIMIT / 8 < FE_WIDE_MAX ? FH_AB_LIMIT / 8 : FE_WIDE_MAX;
static_assert(2 + CO_RENORM * 21 / 20 + 1 <= 8, "running words must stay below 8r between renormalisations");
// The permutation's tables c | m | p | s (contiguous in that order in global memory, load_constants) are staged in LDS by the block
// (reg_tables_to_lds): with one or two waves per SIMD nothing hides a global load, and once the sparse rounds were down to their
// products the kernels ran at the latency of 60 table reads a round instead of at their instruction count.
__host__ __device__ constexpr u32 fh_nrp(int t) { constexpr u32 v[16] = {FH_NRP}; return v[t - 2]; }
template <int T> constexpr bool REG_MF = T >= ZK_FR_MFMA_FROM;             // dense layers on the matrix pipe: M and P are not staged (their fragments pass through LDS an output at a time)
template <int T> constexpr u32 REG_TAB_WORDS = (8 * T + fh_nrp(T) + (REG_MF<T> ? 0 : 2 * T * T + (2 * T - 1) * fh_nrp(T))) * NR;   // matrix pipe: the round constants alone
template <int T> constexpr u32 REG_ABUF_AT = (REG_TAB_WORDS<T> + 3) & ~3u;                                          // 16-byte aligned
template <int T> constexpr u32 REG_LDS_WORDS = REG_ABUF_AT<T> + (REG_MF<T> ? 2 * ((2 * T - 1) * 64 + 4 * T) * 4 : 0);   // + two sparse rounds' blocks (fr_mfma.hip.h; a dense output's t fragments fit)
static_assert(REG_LDS_WORDS<17> * 4 <= 128 * 1024, "the t = 17 tables must leave room in the 160 KiB of LDS");
template <int T>
__device__ __forceinline__ const fe* reg_tables_to_lds(u32* lds) {            // every thread of the block
    const u32* __restrict__ src = (const u32*)g_prm[T - 2].c;
    constexpr u32 NC = (8 * T + fh_nrp(T)) * NR, NM = 2 * T * T * NR, NS = (2 * T - 1) * fh_nrp(T) * NR;
    if constexpr (REG_MF<T>) {
        for (u32 k = threadIdx.x; k < NC; k += blockDim.x) lds[k] = src[k];
    } else {
        for (u32 k = threadIdx.x; k < NC + NM + NS; k += blockDim.x) lds[k] = src[k];
    }
    __syncthreads();
    return (const fe*)lds;
}
template <int T>
__device__ __forceinline__ void poseidon_fr_reg(fe (&st)[T], const fe* __restrict__ tab /* LDS */) {
    constexpr bool MF = REG_MF<T>;
    struct { u32 n_rp; const fe* c; const fe* m; const fe* p; const fe* s; } P;
    P.n_rp = fh_nrp(T); P.c = tab; P.m = tab + 8 * T + fh_nrp(T); P.p = P.m + T * T; P.s = MF ? P.m : P.p + T * T;
    const mf_v4i* mf_m = nullptr; const mf_v4i* mf_p = nullptr; const u64* mf_k = nullptr;
    if constexpr (MF) { mf_m = (const mf_v4i*)g_prm[T - 2].mf_m; mf_p = (const mf_v4i*)g_prm[T - 2].mf_p; mf_k = g_prm[T - 2].mf_k; }
    mf_v4i* const abuf = (mf_v4i*)((u32*)tab + REG_ABUF_AT<T>);
    fh_static_for<0, T>([&](auto I) { st[decltype(I)::value] = fe_add(st[decltype(I)::value], P.c[decltype(I)::value]); });
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 fr = 0; fr < 8; ++fr) {
        if (MF && fr == 4) mf_sparse<T>(st, (const mf_v4i*)g_prm[T - 2].mf_s, (const mf_v4i*)g_prm[T - 2].mf_i, abuf, P.n_rp);
        if (!MF && fr == 4) {
            reg_renorm_tail<T>(st);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (u32 r = 0; r < P.n_rp;) {
                const u32 n = P.n_rp - r < CO_RENORM ? P.n_rp - r : CO_RENORM;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (u32 q = 0; q < n; ++q, ++r) {
                    pow5(st[0]);
                    st[0] = fe_add(st[0], P.c[5 * T + r]);
                    const fe* __restrict__ S = P.s + (size_t)(2 * T - 1) * r;
                    const fe s0 = reg_dot<T, true>(S, 1, st, DOT_SPARSE_REG);
                    fh_static_for<1, T>([&](auto K) { constexpr int k = decltype(K)::value; st[k] = fe_mul_acc(S[T + k - 1], st[0], st[k]); });
                    st[0] = s0;
                }
                reg_renorm_tail<T>(st);
            }
        }
        // S-boxes + round constants of the next linear layer (none after the last S-box layer)
        const fe* __restrict__ c = fr < 3 ? P.c + (fr + 1) * T : fr == 3 ? P.c + 4 * T : P.c + 5 * T + P.n_rp + (fr - 4) * T;
        const bool has_c = fr < 7;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int i = 0; i < T; ++i) {
            fe x = st[0];
            pow5(x);
            if (!MF && has_c) x = fe_add(x, c[i]);                          // matrix pipe: the constants' image rides on the layer's addends
            reg_rotate_in<T>(st, x);
        }
        if constexpr (MF) { mf_dense<T>(st, fr == 3 ? mf_p : mf_m, mf_k + (size_t)fr * T * 8, abuf); continue; }
        const fe* __restrict__ mat = fr == 3 ? P.p : P.m;
        fe out[T];
        fh_static_for<0, T>([&](auto I) { out[decltype(I)::value] = fe_zero(); });
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int i = 0; i < T; ++i) reg_rotate_in<T>(out, reg_dot<T, false>(mat + i, T, st, DOT_DENSE));
        fh_static_for<0, T>([&](auto I) { st[decltype(I)::value] = out[decltype(I)::value]; });
    }
}
// LinearHashBN128::hash_element_array for rows of 5 <= width <= 24 columns: one sponge step of t = NB + 1
#ifndef ZK_LB3
#define ZK_LB3 4
#endif
#ifndef ZK_LB2
#define ZK_LB2 12
#endif
// (waves per SIMD the register budget is held to, measured per block count: three up to 4 blocks -- left alone t = 5 took 254 registers --, two up to 12
// blocks: 2^20 x 24 9.4 -> 7.9 ms, x 33 11.5 -> 10.3; from 13 blocks on two waves spill: x 48 16.8 -> 22.5 ms)
template <int NB>
__global__ __launch_bounds__(256, NB <= ZK_LB3 ? 3 : NB <= ZK_LB2 ? 2 : 1) void bn128_leaf_reg_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests,
                                                                                                     u32 col0 /* first column of this, the row's last, sponge step */) {
    __shared__ __attribute__((aligned(16))) u32 lds[REG_LDS_WORDS<NB + 1>];
    const fe* tab = reg_tables_to_lds<NB + 1>(lds);
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x, i = i0 < height ? i0 : height - 1;   // idle lanes shadow the last row: the matrix pipe wants whole waves
    const u64* __restrict__ v = rows + i * width;
    fe st[NB + 1];
    st[0] = col0 ? load_raw(digests + 4 * i) : fe_zero();                   // the digest of the steps before (bn128_leaf_reg_steps_kernel)
    fh_static_for<0, NB>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const u32 at = col0 + 3 * k, len = width - at < 3 ? width - at : 3;
        st[k + 1] = words_to_fe(v + at, len);
    });
    poseidon_fr_reg<NB + 1>(st, tab);
    if (i0 < height) store_raw(st[FH_OUT_IDX], digests + 4 * i);
}
// rows of more than 48 columns on tall trees: the full sponge steps (16 blocks each, t = 17) of a row one after the other in one lane; the
// last, shorter step is bn128_leaf_reg_kernel<its blocks> on the digest left here.  (Through round 5 these rows took the generic kernel
// with the state in scratch: 2^20 x 49 52.8 ms against 16.8 for 48 columns.)
__global__ __launch_bounds__(256) void bn128_leaf_reg_steps_kernel(const u64* __restrict__ rows, u32 width, u64 height, u32 n_steps, u64* __restrict__ digests) {
    __shared__ __attribute__((aligned(16))) u32 lds[REG_LDS_WORDS<17>];
    const fe* tab = reg_tables_to_lds<17>(lds);
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x, i = i0 < height ? i0 : height - 1;
    const u64* __restrict__ v = rows + i * width;
    fe st[17];
    st[0] = fe_zero();
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 s = 0; s < n_steps; ++s) {
        fh_static_for<0, 16>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const u32 at = 48 * s + 3 * k, len = width - at < 3 ? width - at : 3;
            st[k + 1] = words_to_fe(v + at, len);
        });
        poseidon_fr_reg<17>(st, tab);
        st[0] = st[FH_OUT_IDX];                                             // Poseidon::hash
    }
    if (i0 < height) store_raw(st[0], digests + 4 * i);
}

// hash_node (linearhash_bn128.rs:93-103): parent i = Poseidon(16 digests, init 0)
__global__ __launch_bounds__(64) void bn128_level_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    fe st[17], tmp[17];
    st[0] = fe_zero();
    for (u32 k = 0; k < 16; ++k) st[k + 1] = load_raw(in + (i * 16 + k) * 4);
    poseidon_fr(st, tmp, 17);
    store_raw(st[FH_OUT_IDX], out + 4 * i);
}

// the same with the 17 words in registers (poseidon_fr_reg<17>: no scratch arrays behind run-time indices)
__global__ __launch_bounds__(256) void bn128_level_reg_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32 lds[REG_LDS_WORDS<17>];
    const fe* tab = reg_tables_to_lds<17>(lds);
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x, i = i0 < n_ops ? i0 : n_ops - 1;
    fe st[17];
    st[0] = fe_zero();
    fh_static_for<0, 16>([&](auto K) { st[decltype(K)::value + 1] = load_raw(in + (i * 16 + decltype(K)::value) * 4); });
    poseidon_fr_reg<17>(st, tab);
    if (i0 < n_ops) store_raw(st[FH_OUT_IDX], out + 4 * i);
}

// ---- cooperative permutation for the latency-bound places (small tree levels, few wide rows, the transcript): ONE WAVE per
// permutation.  A one-lane t = 17 permutation is a 1.2 M-instruction dependent chain (4.6 ms); a lone wave is bound by the
// instructions it issues (4.2 cycles each), so the form below is chosen for the fewest instructions per round, not for the
// shortest dependency chain:
//  * dense rounds: lane l = 17 g + i (g < 3) sums a third of column i's products (<= 6: one deferred reduction), the three partial
//    sums meet through LDS -- a column costs 6 multiply passes instead of 17 + 3 reductions + a renormalisation; every replica
//    g holds word i afterwards, the S-boxes run on all of them;
//  * sparse rounds (poseidon_bn128_opt.rs:150-189) unrolled into the recurrence they are.  With y_r = x_r^5 + c_r, w_r = S_r[0..t),
//    v_r[k] = S_r[t + k - 1]:    st_r[k] = st_b[k] + sum_{b <= q < r} v_q[k] y_q,
//    x_{r+1} = w_r[0] y_r + sum_k w_r[k] st_b[k] + sum_{b <= q < r} (sum_k w_r[k] v_q[k]) y_q
//    for any base round b.  Lane 16 + (rho - b) carries x_rho for the rounds of one phase (b < rho <= e), lanes 1..t-1 the words
//    st[k]; each round is x^5 on the wave-uniform x_r, ONE product coef[r][lane] * y_r added into every lane's running value
//    inside the same reduction (fe_mul_acc), and nine v_readlane for x_{r+1} -- against S-box, product, a 32-lane sum and a
//    second product before.  Two phases (at most 47 future rounds fit beside the 16 words); a phase starts by giving the lanes of
//    its rounds sum_k w[k] st_b[k] (16 products).  The coefficient tables are built on the device from S when the constants load.
// Bounds (units of r): dense inputs < 3, thirds < 2, sums < 6 -> S-box (36 <= FH_AB_LIMIT); running values start < 2 and grow by
// < 1.05 a round, renormalised every CO_RENORM = 5 rounds (< 7.3: the S-box's square stays <= 54 <= FH_AB_LIMIT).
constexpr u32 CO_XL0 = 17, CO_XLANES = 47;
static_assert((2 + CO_RENORM * 21 / 20 + 1) * (2 + CO_RENORM * 21 / 20 + 1) <= FH_AB_LIMIT, "running values too large for the S-box square");
__host__ __device__ constexpr u32 co_phase1(u32 n_rp) { return (n_rp + 1) / 2; }   // first round of the second phase; both halves <= CO_XLANES
struct CoopLds { u32 xs[18 * NR]; u32 xp[64 * NR]; };
__device__ __forceinline__ void coop_put(u32* xs, u32 slot, const fe& v) {
#pragma unroll
    for (int k = 0; k < NR; ++k) xs[slot * NR + k] = v.l[k];
}
__device__ __forceinline__ fe coop_get(const u32* xs, u32 j) {
    fe v;
#pragma unroll
    for (int k = 0; k < NR; ++k) v.l[k] = xs[j * NR + k];
    return v;
}
__device__ __forceinline__ fe fe_pick(bool c, const fe& a, const fe& b) {        // limb-wise: a ?: on the structs becomes branches
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) r.l[k] = c ? a.l[k] : b.l[k];
    return r;
}
__device__ __forceinline__ fe wave_bcast(const fe& v, u32 lane) {                 // lane (wave-uniform) -> every lane
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        u32 w = (u32)__builtin_amdgcn_readlane((int)v.l[k], (int)lane);
        asm("" : "+v"(w));   // kept in a vector register on purpose: the compiler would otherwise run the whole S-box of a wave-uniform
        r.l[k] = w;          // x on the scalar unit, four instructions per partial product instead of one
    }
    return r;
}
__device__ __forceinline__ fe load_words(const u32* __restrict__ p) {            // an fe, one limb per load (as a struct it arrives in three pieces that must be repacked -- and waited for -- at once)
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) r.l[k] = p[k];
    return r;
}
__device__ __forceinline__ fe load_lanes(const u32* __restrict__ tab, u32 lane) { // [limb][lane] slice of a table
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) r.l[k] = tab[k * 64 + lane];
    return r;
}
// A lone wave cannot hide a load behind other waves: every table entry a step needs is requested BEFORE the S-box (or the
// product) in front of it, so that the 600-odd instructions of x^5 cover the latency.
struct Cols6 { fe a[6]; };
// the six entries of column i = l % 17 that lane l multiplies: rows 6 g .. 6 g + 5 (rows >= t are met by zero words)
__device__ __forceinline__ void coop_cols_load(Cols6& m, const fe* __restrict__ mat, u32 l, u32 t) {
    const u32 g = l / 17, i = l - 17 * g, ic = i < t ? i : 0, j0 = g * 6;
#pragma unroll
    for (u32 jj = 0; jj < 6; ++jj) {
        const u32 j = j0 + jj;
        m.a[jj] = mat[(size_t)(j < t ? j : t - 1) * t + ic];
    }
}
// st <- (sum_j MAT[j][i] st[j])_i: x = word l of the state in lanes l < t (< 3r); returns word l % 17 in every lane < 51 (< 6r)
__device__ __forceinline__ fe coop_matmul(const Cols6& m, const fe& x, CoopLds& L, u32 l, u32 t) {
    const u32 g = l / 17, i = l - 17 * g, ic = i < t ? i : 0, j0 = g * 6;
    __syncthreads();
    if (l < 18) coop_put(L.xs, l, l < t ? x : fe_zero());             // words t..17 read as zero
    __syncthreads();
    fe_wide w; fe_wide_zero(w);
#pragma unroll
    for (u32 jj = 0; jj < 6; ++jj) {
        const u32 j = j0 + jj < 17 ? j0 + jj : 17;                     // lanes 51..63 (g = 3) and the 18th product of g = 2: times zero
        fe_wide_mac(w, m.a[jj], coop_get(L.xs, j));
    }
    coop_put(L.xp, l, fe_wide_reduce(w));
    __syncthreads();
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) r.l[k] = L.xp[ic * NR + k] + L.xp[(17 + ic) * NR + k] + L.xp[(34 + ic) * NR + k];
    fe_norm_u(r);
    return r;
}
// six of a phase's start-up weights: words j0 .. j0 + 5 of this lane's round (zero table rows beyond t - 1: slot 17 of xs is zero too)
__device__ __forceinline__ void coop_reb_load(Cols6& m, const u32* __restrict__ W, u32 j0, u32 l) {
#pragma unroll
    for (u32 jj = 0; jj < 6; ++jj) {
        const u32 j = j0 + jj < 17 ? j0 + jj : 16;
        m.a[jj] = load_lanes(W + (size_t)(j - 1) * NR * 64, l);
    }
}
__device__ __forceinline__ fe coop_poseidon_fr(fe x, CoopLds& L, u32 t) {   // inlined: as a call it saves and restores 140 registers through scratch
    const u32 l = threadIdx.x & 63, lc = l % 17 < t ? l % 17 : 0;
    const Params P = g_prm[t - 2];
    Cols6 m;
    {
        coop_cols_load(m, P.m, l, t);
        const fe c1 = P.c[t + lc];
        x = fe_add(x, P.c[lc]);
        pow5(x); x = fe_add(x, c1);
        x = coop_matmul(m, x, L, l, t);
    }
#pragma unroll 1
    for (u32 r = 1; r < 4; ++r) {                                       // dense rounds 2..4; the fourth multiplies by P
        coop_cols_load(m, r < 3 ? P.m : P.p, l, t);
        const fe c = P.c[(r + 1) * t + lc];
        pow5(x); x = fe_add(x, c);
        x = coop_matmul(m, x, L, l, t);
    }
    // sparse rounds
    fe xr = wave_bcast(x, 0);                                           // x_0 < 6r
    fe acc = x;                                                         // lanes 1..t-1: st_0[k]
    const fe one = fe_one();
    const u32 n_grp = (t + 4) / 6;                                      // groups of six among the words 1..t-1
#pragma unroll 1
    for (u32 ph = 0; ph < 2; ++ph) {
        const u32 b = ph ? P.b1 : 0, e = ph ? P.n_rp : P.b1;
        const u32* __restrict__ W = P.rebw + (size_t)ph * 16 * NR * 64;
        const u32* __restrict__ C = P.coef + (size_t)b * NR * 64;
        coop_reb_load(m, W, 1, l);
        const u32* __restrict__ RC = (const u32*)(P.c + 5 * t);         // the sparse rounds' constants, limb by limb (wave-uniform)
        fe coef = load_lanes(C, l), cnext = load_words(RC + (size_t)b * NR);
        __syncthreads();
        if (l < 18) coop_put(L.xs, l, l >= 1 && l < t ? acc : fe_zero());   // < 6r entering the first phase, < 2r the second
        __syncthreads();
        {   // lanes of this phase's rounds: sum_k w_{rho-1}[k] st_b[k] (the tables hold zeros for every other lane)
            fe a = fe_zero();
#pragma unroll 1
            for (u32 gi = 0; gi < n_grp; ++gi) {
                const u32 j0 = 1 + 6 * gi;
                Cols6 mn;
                coop_reb_load(mn, W, gi + 1 < n_grp ? j0 + 6 : j0, l);
                fe_wide w; fe_wide_zero(w);
#pragma unroll
                for (u32 jj = 0; jj < 6; ++jj) fe_wide_mac(w, m.a[jj], coop_get(L.xs, j0 + jj < 17 ? j0 + jj : 17));
                a = fe_add(a, fe_wide_reduce(w));
                m = mn;
            }
            if (n_grp > 1) a = fe_mul(a, one);                          // up to three groups: < 6r -> < 2r
            acc = fe_pick(l >= CO_XL0, a, acc);
        }
        // CO_RENORM rounds, then every lane's running value back below 2r.  (An `if` on the round number inside one loop is
        // turned into a product that always runs.)  x_{r+1} is read before that: < 2 + 5 x 1.05 < 7.3r, fine for the S-box.
#pragma unroll 1
        for (u32 r = b; r < e;) {
            const u32 n = e - r < CO_RENORM ? e - r : CO_RENORM;
#pragma unroll 1
            for (u32 k = 0; k < n; ++k, ++r) {
                const fe cf = coef, c = cnext;
                const u32 rn = r + 1 < e ? r + 1 : r;                   // a round ahead; the last one re-reads its own
                coef = load_lanes(P.coef + (size_t)rn * NR * 64, l);
                cnext = load_words(RC + (size_t)rn * NR);
                fe y = xr;
                pow5(y);
#pragma unroll
                for (int q = 0; q < NR; ++q) y.l[q] += c.l[q];          // limbs < 2^30: fe_mul_acc takes them as they are
                acc = fe_mul_acc(cf, y, acc);
                xr = wave_bcast(acc, CO_XL0 + (r - b));                 // x_{r+1}
            }
            acc = fe_mul(acc, one);
        }
    }
    coop_cols_load(m, P.m, l, t);
    x = fe_pick(l == 0, xr, acc);                                       // word layout again: x < 7.3r, st[k] < 2r
#pragma unroll 1
    for (u32 r = 0; r < 4; ++r) {
        const fe c = r < 3 ? P.c[5 * t + P.n_rp + r * t + lc] : fe_zero();
        if (r) coop_cols_load(m, P.m, l, t);
        pow5(x); x = fe_add(x, c);
        x = coop_matmul(m, x, L, l, t);
    }
    return fe_mul(x, one);                                              // < 2r for the callers (digests chained into the next sponge step)
}
// the sparse rounds' cooperative tables for one t (see above); one thread per (round, lane) and per (phase, word, lane)
__global__ void coop_tables_kernel(Params P, u32* __restrict__ coef, u32* __restrict__ rebw) {
    const u32 t = P.t, n_rp = P.n_rp, b1 = co_phase1(n_rp);
    const u32 id = blockIdx.x * blockDim.x + threadIdx.x, lane = id & 63, idx = id >> 6;
    const fe one = fe_one();
    if (idx < n_rp) {
        const u32 r = idx, b = r >= b1 ? b1 : 0, e = r >= b1 ? n_rp : b1;
        const fe* __restrict__ Sr = P.s + (size_t)(2 * t - 1) * r;
        fe v = fe_zero();
        if (lane >= 1 && lane < t) v = Sr[t + lane - 1];
        else if (lane >= CO_XL0 && b + (lane - 16) <= e) {
            const u32 rho = b + (lane - 16);
            const fe* __restrict__ Sw = P.s + (size_t)(2 * t - 1) * (rho - 1);
            if (r == rho - 1) v = Sw[0];
            else if (r < rho - 1) {
                for (u32 j = 1; j < t; ++j) v = fe_mul(fe_add(v, fe_mul(Sw[j], Sr[t + j - 1])), one);   // < 4r -> < 2r
            }
        }
        v = fe_canon(fe_mul(v, one));
        for (int k = 0; k < NR; ++k) coef[((size_t)r * NR + k) * 64 + lane] = v.l[k];
    } else if (idx < n_rp + 32) {
        const u32 q = idx - n_rp, ph = q >> 4, j = (q & 15) + 1, b = ph ? b1 : 0, e = ph ? n_rp : b1;
        fe v = fe_zero();
        if (j < t && lane >= CO_XL0 && b + (lane - 16) <= e) v = P.s[(size_t)(2 * t - 1) * (b + (lane - 16) - 1) + j];
        for (int k = 0; k < NR; ++k) rebw[((size_t)(ph * 16 + (j - 1)) * NR + k) * 64 + lane] = v.l[k];
    }
}
__global__ __launch_bounds__(64) void bn128_level_coop_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    __shared__ CoopLds L;
    const u32 l = threadIdx.x;
    const u64 i = blockIdx.x;
    fe x = fe_zero();
    if (l >= 1 && l <= 16) x = load_raw(in + (i * 16 + (l - 1)) * 4);
    x = coop_poseidon_fr(x, L, 17);
    if (l == FH_OUT_IDX) store_raw(x, out + 4 * i);
}
// LinearHash of a few, possibly very wide rows (the FRI trees of a large fold: final.starkStruct.*.json commits 2^7 rows of 3072
// words): one lane per row would run 64 sponge steps of 1.2 M instructions each on 128 lanes.  One wave per row instead, the
// sponge steps of linearhash_bn128.rs:105-131 in order, the digest handed from lane FH_OUT_IDX to lane 0.
__global__ __launch_bounds__(64) void bn128_leaf_coop_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    __shared__ CoopLds L;
    const u32 l = threadIdx.x;
    const u64 i = blockIdx.x;
    const u64* __restrict__ v = rows + i * width;
    const u32 nb = (width - 1) / 3 + 1;
    fe digest = fe_zero();
    for (u32 b = 0; b < nb; b += 16) {
        const u32 sz = nb - b < 16 ? nb - b : 16;
        fe x = fe_zero();
        if (l == 0) x = digest;
        else if (l <= sz) {
            const u32 at = 3 * (b + l - 1), len = width - at < 3 ? width - at : 3;
            x = words_to_fe(v + at, len);
        }
        x = coop_poseidon_fr(x, L, sz + 1);
        digest = wave_bcast(x, FH_OUT_IDX);
    }
    if (l == 0) store_raw(digest, digests + 4 * i);
}
// Poseidon::hash_ex for a handful of permutations (the transcript: one sponge step at a time): one wave each
__global__ __launch_bounds__(64) void bn128_poseidon_coop_kernel(const u64* __restrict__ inp, u64 n, u32 n_in, const u64* __restrict__ init,
                                                                 u32 n_out, u64* __restrict__ out) {
    __shared__ CoopLds L;
    const u32 l = threadIdx.x;
    const u64 i = blockIdx.x;
    fe x = fe_zero();
    if (l == 0) x = load_raw(init);
    else if (l <= n_in) x = load_raw(inp + (i * n_in + (l - 1)) * 4);
    x = coop_poseidon_fr(x, L, n_in + 1);
    if (l < n_out) store_raw(x, out + (i * n_out + l) * 4);
}

// ---- L lanes per permutation, for tree levels of a few thousand parents: too many for one wave each (every SIMD would run
// ---- several 153 us permutations one after the other), too few for one lane each (a t = 17 permutation is a 2.5 ms chain and
// ---- 8192 of them fill an eighth of the SIMDs).  Lane q of a group keeps the words q, q + L, q + 2L (S = ceil(17 / L) slots) in
// ---- registers; words meet through LDS for the dense products, the matrices M and P are staged in LDS once per block.  A sparse
// ---- round: x broadcast to the group, x^5 in every lane, each lane's S products of the row, reduced, canonical, summed over the
// ---- group (< L r: the next S-box takes it as it is for L <= 8), and S fused product-accumulates for the column.
// ---- Bounds as in the cooperative form: running words grow by < 1.05 r a round and are renormalised every CO_RENORM rounds.
template <int L, int S>
__device__ __forceinline__ void grp_matmul(fe (&st)[S], const u32 (&wd)[S], const u32* __restrict__ mat /* LDS */, u32* __restrict__ xg /* LDS, this group's words */, u32 q) {
    constexpr int T = 17;                                                        // st < 3r in, < 6r out
    __syncthreads();
    fh_static_for<0, S>([&](auto SI) { constexpr int sl = decltype(SI)::value; if (sl * L + q < (u32)T) coop_put(xg, sl * L + q, st[sl]); });
    __syncthreads();
    fh_static_for<0, S>([&](auto SI) { st[decltype(SI)::value] = fe_zero(); });
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 j0 = 0; j0 < 18; j0 += 6) {
        fe_wide w[S];
        fh_static_for<0, S>([&](auto SI) { fe_wide_zero(w[decltype(SI)::value]); });
#pragma unroll 2
        for (u32 jj = 0; jj < 6; ++jj) {
            const u32 j = j0 + jj;
            if (j < (u32)T) {
                const fe xj = coop_get(xg, j);
                fh_static_for<0, S>([&](auto SI) { constexpr int sl = decltype(SI)::value; fe_wide_mac(w[sl], coop_get(mat, j * T + wd[sl]), xj); });
            }
        }
        fh_static_for<0, S>([&](auto SI) { constexpr int sl = decltype(SI)::value; st[sl] = fe_add(st[sl], fe_wide_reduce(w[sl])); });
    }
}
template <int S>
__device__ __forceinline__ void grp_sbox_layer(fe (&st)[S], const u32 (&wd)[S], const fe* __restrict__ c) {   // x^5 + the next constants (c may be null)
    fe cc[S];
    fh_static_for<0, S>([&](auto SI) { constexpr int sl = decltype(SI)::value; cc[sl] = c ? c[wd[sl]] : fe_zero(); });
    fh_static_for<0, S>([&](auto SI) { constexpr int sl = decltype(SI)::value; pow5(st[sl]); st[sl] = fe_add(st[sl], cc[sl]); });
}
template <int L>
__global__ __launch_bounds__(64) void bn128_level_grp_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    constexpr int T = 17, S = (T + L - 1) / L, G = 64 / L;
    static_assert(L == 8 || L == 4, "group sums below are bounded for L <= 8");
    static_assert((u32)L * (u32)L <= FH_AB_LIMIT, "a sum of L canonical values must fit the S-box square");
    __shared__ u32 mats[2][T * T * NR];
    __shared__ u32 xs[G][T * NR];
    const Params P = g_prm[T - 2];
    const u32 q = threadIdx.x % L, g = threadIdx.x / L;
    {   // both matrices into LDS (289 x 9 words each)
        const u32* __restrict__ m0 = (const u32*)P.m; const u32* __restrict__ m1 = (const u32*)P.p;
        for (u32 k = threadIdx.x; k < T * T * NR; k += 64) { mats[0][k] = m0[k]; mats[1][k] = m1[k]; }
    }
    const u64 i = (u64)blockIdx.x * G + g, ic = i < n_ops ? i : n_ops - 1;      // an idle group shadows the last parent
    const fe one = fe_one();
    fe st[S];
    u32 wd[S];                                                                  // this lane's words; slots past the state work on word 0's constants and are never read back
    fh_static_for<0, S>([&](auto SI) {
        constexpr int sl = decltype(SI)::value;
        const u32 w = sl * L + q;
        wd[sl] = w < T ? w : 0;
        st[sl] = fe_zero();
        if (w >= 1 && w < T) st[sl] = load_raw(in + (ic * 16 + (w - 1)) * 4);
        st[sl] = fe_add(st[sl], P.c[wd[sl]]);
    });
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 r = 0; r < 4; ++r) {
        grp_sbox_layer<S>(st, wd, P.c + (r + 1) * T);
        grp_matmul<L, S>(st, wd, r < 3 ? mats[0] : mats[1], xs[g], q);
    }
    // sparse rounds: word 0 lives in slot 0 of lane 0 of the group.  CO_RENORM rounds, then every running word back below 2r.
    const u32 src0 = threadIdx.x - q;
    const bool is0 = q == 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 r = 0; r < P.n_rp;) {
        const u32 n = P.n_rp - r < CO_RENORM ? P.n_rp - r : CO_RENORM;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (u32 k = 0; k < n; ++k, ++r) {
            const fe* __restrict__ Sr = P.s + (size_t)(2 * T - 1) * r;
            fe wv[S], vv[S];
            fh_static_for<0, S>([&](auto SI) {                                  // requested in front of the S-box
                constexpr int sl = decltype(SI)::value;
                wv[sl] = load_words((const u32*)(Sr + wd[sl]));
                vv[sl] = load_words((const u32*)(Sr + T + (wd[sl] ? wd[sl] : 1) - 1));
            });
            const fe c = load_words((const u32*)(P.c + 5 * T + r));
            fe y;
#pragma unroll
            for (int e = 0; e < NR; ++e) y.l[e] = (u32)__shfl((int)st[0].l[e], (int)src0, 64);
            pow5(y);
#pragma unroll
            for (int e = 0; e < NR; ++e) y.l[e] += c.l[e];                      // limbs < 2^30, value < 3r
            fe_wide w; fe_wide_zero(w);
            fh_static_for<0, S>([&](auto SI) {
                constexpr int sl = decltype(SI)::value;
                const bool live = sl * L + q < (u32)T;
                fe_wide_mac(w, fe_pick(live, wv[sl], fe_zero()), fe_pick(sl == 0 && is0, y, st[sl]));   // <= 3 products of < 7.3r (word 0: < 3r) by < r
            });
            fe p = fe_canon(fe_wide_reduce(w));
#pragma unroll
            for (int m = 1; m < L; m <<= 1) {
                fe o;
#pragma unroll
                for (int e = 0; e < NR; ++e) o.l[e] = (u32)__shfl_xor((int)p.l[e], m, 64);
#pragma unroll
                for (int e = 0; e < NR; ++e) p.l[e] += o.l[e];
                if (m == 2 || m * 2 >= L) fe_norm_u(p);                         // at most four summands per limb between two carry passes
            }
            fh_static_for<0, S>([&](auto SI) {
                constexpr int sl = decltype(SI)::value;
                const fe upd = fe_mul_acc(vv[sl], y, st[sl]);
                st[sl] = fe_pick(sl == 0 && is0, p, upd);                       // word 0 <- the row's sum (< L r)
            });
        }
        fh_static_for<0, S>([&](auto SI) { constexpr int sl = decltype(SI)::value; st[sl] = fe_pick(sl == 0 && is0, st[sl], fe_mul(st[sl], one)); });
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 r = 0; r < 4; ++r) {
        grp_sbox_layer<S>(st, wd, r < 3 ? P.c + 5 * T + P.n_rp + r * T : nullptr);
        grp_matmul<L, S>(st, wd, mats[0], xs[g], q);
    }
    if (i < n_ops && q == FH_OUT_IDX) store_raw(st[0], out + 4 * i);
}

struct DeviceTables { fe* all = nullptr; bool ready = false; };
DeviceTables g_tables[64];
const u32 NRP[16] = {FH_NRP};   // = fh_nrp(t) for t = 2..17

// The matrix-pipe tables of one t (fr_mfma.hip.h): fragments of M and P, and per dense layer the addends carrying the image of the round
// constants that follow the layer's S-boxes (poseidon_fr_reg's order: layers 0..2 M, 3 P, 4..7 M; none after the last S-boxes).
void mf_upload(const unsigned char* canon, size_t off_c, size_t off_m, size_t off_p, size_t off_s, int T, u32 n_rp, Params& prm) {
    const MfInt q = mf_modulus();
    const size_t fb = (size_t)T * T * 1024;
    std::vector<signed char> frag(2 * fb);
    std::vector<MfInt> corr(2 * T);
    std::string err = mf_build_matrix(canon + 32 * off_m, T, frag.data(), corr.data());
    if (err.empty()) err = mf_build_matrix(canon + 32 * off_p, T, frag.data() + fb, corr.data() + T);
    if (!err.empty()) throw Error(std::string(FH_NAME " Poseidon, matrix-pipe tables of t = ") + std::to_string(T) + ": " + err);
    std::vector<u64> K((size_t)8 * T * 8);
    std::vector<MfInt> cm(T);
    for (int layer = 0; layer < 8; ++layer) {
        const bool isp = layer == 3, has_c = layer < 7;
        const size_t c_at = layer < 3 ? (size_t)(layer + 1) * T : layer == 3 ? (size_t)4 * T : (size_t)5 * T + n_rp + (size_t)(layer - 4) * T;
        if (has_c) for (int j = 0; j < T; ++j) cm[j] = mf_shlmod(mf_from_bytes(canon + 32 * (off_c + c_at + j)), NR * LB + LB, q);   // internal form, times the step's 2^29
        const unsigned char* mat = canon + 32 * (isp ? off_p : off_m);
        for (int o = 0; o < T; ++o) {
            MfInt add = mf_zero();
            if (has_c) for (int j = 0; j < T; ++j) add = mf_addmod(add, mf_mulmod(cm[j], mf_from_bytes(mat + 32 * ((size_t)j * T + o)), q), q);
            mf_addends(add, corr[(isp ? T : 0) + o], &K[((size_t)layer * T + o) * 8]);
        }
    }
    const size_t sb = mf_sparse_round_bytes(T) * n_rp;
    std::vector<unsigned char> sp(sb + 1024);
    err = mf_build_sparse(canon + 32 * off_s, canon + 32 * (off_c + (size_t)5 * T), T, (int)n_rp, sp.data(), (signed char*)sp.data() + sb);
    if (!err.empty()) throw Error(std::string(FH_NAME " Poseidon, matrix-pipe tables of t = ") + std::to_string(T) + ": " + err);
    void* d_sp = nullptr;
    ZK_HIP(hipMalloc(&d_sp, sp.size()));
    ZK_HIP(hipMemcpy(d_sp, sp.data(), sp.size(), hipMemcpyHostToDevice));
    prm.mf_s = d_sp; prm.mf_i = (const char*)d_sp + sb;
    void* d_frag = nullptr; u64* d_k = nullptr;
    ZK_HIP(hipMalloc(&d_frag, 2 * fb));
    ZK_HIP(hipMalloc((void**)&d_k, K.size() * 8));
    ZK_HIP(hipMemcpy(d_frag, frag.data(), 2 * fb, hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(d_k, K.data(), K.size() * 8, hipMemcpyHostToDevice));
    prm.mf_m = d_frag; prm.mf_p = (const char*)d_frag + fb; prm.mf_k = d_k;
}

void require_tables() {
    int dev; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64 && g_tables[dev].ready, FH_NAME " Poseidon constants not loaded (zk_" FH_NAME "_load_constants)");
}

}  // namespace

// file format: tools/gen_poseidon_bn128_constants.py
namespace {
struct TabOff { size_t c, m, p, s; u32 t; };
size_t read_constants(const char* path, std::vector<unsigned char>& canon /* all 32-byte values back to back */, TabOff (&off)[16]) {
    FILE* f = fopen(path, "rb");
    if (!f) throw Error(std::string("cannot open ") + path);
    std::vector<unsigned char> buf;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    const size_t got = fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    ZK_REQUIRE(got == buf.size() && buf.size() >= 8 && !memcmp(buf.data(), "PBN1", 4), "bad Poseidon constants file");
    uint32_t nt; memcpy(&nt, buf.data() + 4, 4);
    ZK_REQUIRE(nt == 16, "bad Poseidon constants file (t range)");
    size_t pos = 8, count = 0;
    for (int k = 0; k < 16; ++k) {
        ZK_REQUIRE(pos + 12 <= buf.size(), "truncated Poseidon constants file");
        uint32_t h[3]; memcpy(h, buf.data() + pos, 12); pos += 12;
        const u32 t = h[0], n_c = h[1], n_s = h[2];
        ZK_REQUIRE(t == (u32)k + 2 && n_c == 5 * t + NRP[k] + 3 * t && n_s == (2 * t - 1) * NRP[k], "unexpected Poseidon table shape");
        const size_t n = n_c + 2 * (size_t)t * t + n_s;
        ZK_REQUIRE(pos + 32 * n <= buf.size(), "truncated Poseidon constants file");
        off[k] = {count, count + n_c, count + n_c + (size_t)t * t, count + n_c + 2 * (size_t)t * t, t};
        canon.insert(canon.end(), buf.begin() + pos, buf.begin() + pos + 32 * n);
        pos += 32 * n; count += n;
    }
    return count;
}
}
// host arithmetic only (no device): the matrix-pipe tables of every t against the constants they are built from; "" or what is wrong
std::string FH_FN(tables_selfcheck)(const char* path) {
    std::vector<unsigned char> canon; TabOff off[16];
    read_constants(path, canon, off);
    for (int k = 0; k < 16; ++k) {
        if ((int)off[k].t < ZK_FR_MFMA_FROM) continue;
        for (int which = 0; which < 2; ++which) {
            const std::string why = mf_selfcheck(canon.data() + 32 * (which ? off[k].p : off[k].m), (int)off[k].t, 977 * k + which);
            if (!why.empty()) return std::string(FH_NAME " Poseidon, matrix ") + (which ? "P" : "M") + ": " + why;
        }
        const std::string why = mf_selfcheck_sparse(canon.data() + 32 * off[k].s, canon.data() + 32 * (off[k].c + (size_t)5 * off[k].t), (int)off[k].t, (int)NRP[k], 31 * k);
        if (!why.empty()) return std::string(FH_NAME " Poseidon: ") + why;
    }
    return "";
}
void FH_FN(load_constants)(const char* path) {
    int dev; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    if (g_tables[dev].ready) return;
    std::vector<unsigned char> canon; TabOff off[16];
    const size_t count = read_constants(path, canon, off);
    DevBuf d_canon; d_canon.reserve(canon.size());
    ZK_HIP(hipMemcpy(d_canon.p, canon.data(), canon.size(), hipMemcpyHostToDevice));
    fe* d_all = nullptr;
    ZK_HIP(hipMalloc((void**)&d_all, count * sizeof(fe)));
    hipLaunchKernelGGL(bn128_convert_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, (const u32*)d_canon.p, (u64)count, d_all);
    ZK_HIP(hipGetLastError());
    Params prm[16];
    size_t n_coop = 0;
    for (int k = 0; k < 16; ++k) n_coop += ((size_t)NRP[k] + 32) * NR * 64;
    u32* d_coop = nullptr;
    ZK_HIP(hipMalloc((void**)&d_coop, n_coop * sizeof(u32)));
    size_t at = 0;
    for (int k = 0; k < 16; ++k) {
        ZK_REQUIRE(co_phase1(NRP[k]) <= CO_XLANES && NRP[k] - co_phase1(NRP[k]) <= CO_XLANES, "too many sparse rounds for the cooperative form");
        u32* coef = d_coop + at; u32* rebw = coef + (size_t)NRP[k] * NR * 64;
        at += ((size_t)NRP[k] + 32) * NR * 64;
        prm[k] = {off[k].t, NRP[k], d_all + off[k].c, d_all + off[k].m, d_all + off[k].p, d_all + off[k].s, coef, rebw, co_phase1(NRP[k]), nullptr, nullptr, nullptr, nullptr, nullptr};
        if ((int)off[k].t >= ZK_FR_MFMA_FROM) mf_upload(canon.data(), off[k].c, off[k].m, off[k].p, off[k].s, (int)off[k].t, NRP[k], prm[k]);
        hipLaunchKernelGGL(coop_tables_kernel, dim3(NRP[k] + 32), dim3(64), 0, nullptr, prm[k], coef, rebw);
        ZK_HIP(hipGetLastError());
    }
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_prm), prm, sizeof(prm)));
    ZK_HIP(hipDeviceSynchronize());
    g_tables[dev].all = d_all; g_tables[dev].ready = true;
}

void FH_FN(poseidon_dev)(const u64* d_inp, uint64_t n, uint32_t n_in, const u64* d_init, uint32_t n_out, u64* d_out, hipStream_t st) {
    require_tables();
    ZK_REQUIRE(n_in >= 1 && n_in <= 16, "Wrong inputs length");            // poseidon_bn128_opt.rs:99-105
    ZK_REQUIRE(n_out >= 1 && n_out <= n_in + 1, "Wrong output length");
    if (n == 0) return;
    if (n <= 4096)  // latency-bound
        hipLaunchKernelGGL(bn128_poseidon_coop_kernel, dim3((unsigned)n), dim3(64), 0, st, d_inp, n, n_in, d_init, n_out, d_out);
    else
        hipLaunchKernelGGL(bn128_poseidon_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_inp, n, n_in, d_init, n_out, d_out);
    ZK_HIP(hipGetLastError());
}

uint64_t FH_FN(merkle_n_nodes)(uint64_t n_) {  // merklehash_bn128.rs:26-39
    uint64_t n = n_, next_n = (n - 1) / 16 + 1, acc = next_n * 16;
    while (n > 1) {
        n = next_n; next_n = (n - 1) / 16 + 1;
        if (n > 1) acc += next_n * 16; else acc += 1;
    }
    return acc;
}

void FH_FN(linearhash_rows_dev)(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st) {
    require_tables();
    if (height == 0) return;
    const dim3 grid((unsigned)((height + 63) / 64)), blk(64), grid_reg((unsigned)((height + 255) / 256));
    const u32 nb = width ? (width - 1) / 3 + 1 : 0;
    if (width > 4 && height <= 4096) {   // latency-bound: 32 lanes per row
        hipLaunchKernelGGL(bn128_leaf_coop_kernel, dim3((unsigned)height), blk, 0, st, d_rows, width, height, d_digests);
        ZK_HIP(hipGetLastError());
        return;
    }
    static const int reg_max = getenv("ZK_FRHASH_REG_MAX") ? atoi(getenv("ZK_FRHASH_REG_MAX")) : 1 << 30;   // tuning knob: largest block count that takes the register kernels
    if (width <= 4 || (int)nb > reg_max) {   // width <= 4: no hash
        hipLaunchKernelGGL(bn128_leaf_kernel, grid, blk, 0, st, d_rows, width, height, d_digests);
        ZK_HIP(hipGetLastError());
        return;
    }
    // one sponge step with the state in registers; rows of more than 16 blocks: their full steps first, the digest handed on through d_digests
    const u32 n_steps = (nb - 1) / 16, last = nb - 16 * n_steps, col0 = 48 * n_steps;       // last = 1..16 blocks
    if (n_steps) {
        hipLaunchKernelGGL(bn128_leaf_reg_steps_kernel, grid_reg, dim3(256), 0, st, d_rows, width, height, n_steps, d_digests);
        ZK_HIP(hipGetLastError());
    }
#define ZK_LEAF_REG(N) case N: hipLaunchKernelGGL(bn128_leaf_reg_kernel<N>, grid_reg, dim3(256), 0, st, d_rows, width, height, d_digests, col0); break;
    switch (last) {
        ZK_LEAF_REG(1) ZK_LEAF_REG(2) ZK_LEAF_REG(3) ZK_LEAF_REG(4) ZK_LEAF_REG(5) ZK_LEAF_REG(6) ZK_LEAF_REG(7) ZK_LEAF_REG(8)
        ZK_LEAF_REG(9) ZK_LEAF_REG(10) ZK_LEAF_REG(11) ZK_LEAF_REG(12) ZK_LEAF_REG(13) ZK_LEAF_REG(14) ZK_LEAF_REG(15) ZK_LEAF_REG(16)
    }
#undef ZK_LEAF_REG
    ZK_HIP(hipGetLastError());
}

// nodes: FH_FN(merkle_n_nodes)(height) * 4 words, zero-filled by this call (merklehash_bn128.rs:196-239)
void FH_FN(merkelize_dev)(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st) {
    require_tables();
    ZK_REQUIRE(height >= 1, "merkelize: height must be >= 1");
    const uint64_t nn = FH_FN(merkle_n_nodes)(height);
    ZK_HIP(hipMemsetAsync(d_nodes, 0, nn * 32, st));
    if (width) FH_FN(linearhash_rows_dev)(d_rows, width, height, d_nodes, st);
    uint64_t n = height, next = (n - 1) / 16 + 1, p_in = 0, p_out = next * 16;
    while (n > 1) {
        static const bool level_reg = !getenv("ZK_FR_LEVEL_REG") || atoi(getenv("ZK_FR_LEVEL_REG"));   // tuning knob: 0 = the generic one-lane kernel
        static const u64 coop_upto = getenv("ZK_FR_LEVEL_COOP") ? strtoull(getenv("ZK_FR_LEVEL_COOP"), nullptr, 10) : 16384;
        static const u64 grp_from = getenv("ZK_FR_LEVEL_GRP_FROM") ? strtoull(getenv("ZK_FR_LEVEL_GRP_FROM"), nullptr, 10) : 2561;    // tuning knobs: parents from .. upto take
        static const u64 grp_upto = getenv("ZK_FR_LEVEL_GRP_UPTO") ? strtoull(getenv("ZK_FR_LEVEL_GRP_UPTO"), nullptr, 10) : 16384;   // eight lanes each (0 upto = never); round 6: 32768 -> 16384, the one-lane kernel on the matrix pipe takes 32 768 parents in 1.0 ms against 1.9
        if (next >= grp_from && next <= grp_upto)
            hipLaunchKernelGGL(bn128_level_grp_kernel<8>, dim3((unsigned)((next + 7) / 8)), dim3(64), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        else if (next <= coop_upto)  // latency-bound: one wave per parent
            hipLaunchKernelGGL(bn128_level_coop_kernel, dim3((unsigned)next), dim3(64), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        else if (level_reg)
            hipLaunchKernelGGL(bn128_level_reg_kernel, dim3((unsigned)((next + 255) / 256)), dim3(256), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        else
            hipLaunchKernelGGL(bn128_level_kernel, dim3((unsigned)((next + 63) / 64)), dim3(64), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        ZK_HIP(hipGetLastError());
        n = next; next = (n - 1) / 16 + 1; p_in = p_out; p_out = p_in + next * 16;
    }
}

