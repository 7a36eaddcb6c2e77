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
struct Params { u32 t, n_rp; const fe* c; const fe* m; const fe* p; const fe* s; };
__device__ Params g_prm[16];

__device__ __forceinline__ void pow5(fe& x) { const fe x2 = fe_sqr(x), x4 = fe_sqr(x2); x = fe_mul(x4, x); }  // poseidon_bn128_opt.rs:88-94

// Sums of products with deferred reduction (fe29_impl.hip.h fe_wide): the tables are canonical (< r), so a group of g
// products with operands < B r needs g B <= FH_AB_LIMIT (168 for BN254's r, 68 for BLS12-381's).
constexpr u32 DOT_DENSE = FE_WIDE_MAX;                                       // dense products: state < 3r (6 x 3 = 18)
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
template <int T>
__device__ __forceinline__ void reg_rotate_in(fe (&a)[T], const fe& last) {
    fh_static_for<0, T - 1>([&](auto I) { a[decltype(I)::value] = a[decltype(I)::value + 1]; });
    a[T - 1] = last;
}
template <int T>
__device__ __forceinline__ fe reg_dot(const fe* __restrict__ a, u32 stride, const fe (&x)[T], u32 grp) {
    fe_wide w; fe_wide_zero(w);
    fe acc = fe_zero();
    u32 cnt = 0;
    fh_static_for<0, T>([&](auto J) {
        constexpr int j = decltype(J)::value;
        fe_wide_mac(w, a[(size_t)j * stride], x[j]);
        if (++cnt == grp || j + 1 == T) { acc = fe_add(acc, fe_wide_reduce(w)); cnt = 0; fe_wide_zero(w); }
    });
    return (u32)T > grp ? fe_renorm(acc) : acc;
}
template <int T>
__device__ __forceinline__ void poseidon_fr_reg(fe (&st)[T]) {
    const Params P = g_prm[T - 2];
    fh_static_for<0, T>([&](auto I) { st[decltype(I)::value] = fe_add(st[decltype(I)::value], P.c[decltype(I)::value]); });
#pragma unroll 1
    for (u32 fr = 0; fr < 8; ++fr) {
        if (fr == 4) {
#pragma unroll 1
            for (u32 r = 0; r < P.n_rp; ++r) {
                pow5(st[0]);
                st[0] = fe_add(st[0], P.c[5 * T + r]);
                const fe* __restrict__ S = P.s + (size_t)(2 * T - 1) * r;
                const fe s0 = reg_dot<T>(S, 1, st, DOT_SPARSE);
                const bool renorm = r % PR_RENORM == PR_RENORM - 1;
#pragma unroll 1
                for (int k = 1; k < T; ++k) {                       // word 1 is updated, then the tail st[1..T) rotates
                    fe v = fe_add(st[1], fe_mul(S[T + k - 1], st[0]));
                    if (renorm) v = fe_renorm(v);
                    fh_static_for<1, T - 1>([&](auto I) { st[decltype(I)::value] = st[decltype(I)::value + 1]; });
                    st[T - 1] = v;
                }
                st[0] = s0;
            }
#pragma unroll 1
            for (int k = 1; k < T; ++k) {
                const fe v = fe_renorm(st[1]);
                fh_static_for<1, T - 1>([&](auto I) { st[decltype(I)::value] = st[decltype(I)::value + 1]; });
                st[T - 1] = v;
            }
        }
        // S-boxes + round constants of the next linear layer (none after the last S-box layer)
        const fe* __restrict__ c = fr < 3 ? P.c + (fr + 1) * T : fr == 3 ? P.c + 4 * T : P.c + 5 * T + P.n_rp + (fr - 4) * T;
        const bool has_c = fr < 7;
#pragma unroll 1
        for (int i = 0; i < T; ++i) {
            fe x = st[0];
            pow5(x);
            if (has_c) x = fe_add(x, c[i]);
            reg_rotate_in<T>(st, x);
        }
        const fe* __restrict__ mat = fr == 3 ? P.p : P.m;
        fe out[T];
        fh_static_for<0, T>([&](auto I) { out[decltype(I)::value] = fe_zero(); });
#pragma unroll 1
        for (int i = 0; i < T; ++i) reg_rotate_in<T>(out, reg_dot<T>(mat + i, T, st, DOT_DENSE));
        fh_static_for<0, T>([&](auto I) { st[decltype(I)::value] = out[decltype(I)::value]; });
    }
}
// LinearHashBN128::hash_element_array for rows of 5 <= width <= 24 columns: one sponge step of t = NB + 1
template <int NB>
__global__ __launch_bounds__(64) void bn128_leaf_reg_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= height) return;
    const u64* __restrict__ v = rows + i * width;
    fe st[NB + 1];
    st[0] = fe_zero();
    fh_static_for<0, NB>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const u32 at = 3 * k, len = width - at < 3 ? width - at : 3;
        st[k + 1] = words_to_fe(v + at, len);
    });
    poseidon_fr_reg<NB + 1>(st);
    store_raw(st[FH_OUT_IDX], digests + 4 * i);
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


// ---- cooperative permutation for the small levels of a tree: 32 lanes per permutation, lane l < t owns st[l].
// A one-lane t = 17 permutation is a 1.2 M-instruction dependent chain (4.6 ms): the last three levels of every
// tree (1 + 16 + 256 parents) would cost 14 ms.  Here the 17 columns of a dense product, the 17 S-boxes and the
// 16 column updates of a sparse round run side by side; words are exchanged through LDS (`xs`: 17 x 9 words per
// group).  Same bounds as poseidon_fr.  Called by all 64 threads of a block (two groups), uniform control flow.
__device__ __forceinline__ void coop_put(u32* xs, int l, u32 t, const fe& v) {
    if ((u32)l < t) {
#pragma unroll
        for (int k = 0; k < NR; ++k) xs[l * NR + k] = v.l[k];
    }
}
__device__ __forceinline__ fe coop_get(const u32* xs, u32 j) {
    fe v;
#pragma unroll
    for (int k = 0; k < NR; ++k) v.l[k] = xs[j * NR + k];
    return v;
}
__device__ fe coop_matmul(const fe* __restrict__ mat, const fe& x, u32* xs, int l, u32 t) {
    __syncthreads();
    coop_put(xs, l, t, x);
    __syncthreads();
    const u32 lc = (u32)l < t ? l : 0;
    return dot_products(mat + lc, t, [&](u32 j) { return coop_get(xs, j); }, t, DOT_DENSE);
}
// Partial rounds exchange nothing through LDS.  A block is one wave holding two groups of 32 lanes (two DPP rows each): lane 0's
// S-box output reaches its group through v_readlane; the t products S[j] st[j] are summed towards lane 0 by shifted row
// additions (limbs are 29 bits wide: four summands fit a word, then a carry pass) and one v_readlane across the two rows --
// instead of seventeen LDS round trips walked by lane 0 alone while the other lanes wait, with four barriers a round.
__device__ __forceinline__ fe fe_pick(bool c, const fe& a, const fe& b) {        // limb-wise: a ?: on the structs becomes branches
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) r.l[k] = c ? a.l[k] : b.l[k];
    return r;
}
__device__ __forceinline__ fe group_bcast0(const fe& v, int g) {                 // lane 0 of each 32-lane group -> its group
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)v.l[k], 0), hi = (u32)__builtin_amdgcn_readlane((int)v.l[k], 32);
        r.l[k] = g ? hi : lo;
    }
    return r;
}
static_assert(NR == 9, "the row additions below name nine limbs");
#define ZK_FR_ROW_ADD_SHL(N)                                                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                                 \
                 "v_add_u32_dpp %0, %0, %0 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %1, %1, %1 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %2, %2, %2 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %3, %3, %3 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %4, %4, %4 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %5, %5, %5 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %6, %6, %6 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %7, %7, %7 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
                 "v_add_u32_dpp %8, %8, %8 row_shl:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                    \
                 : "+v"(v.l[0]), "+v"(v.l[1]), "+v"(v.l[2]), "+v"(v.l[3]), "+v"(v.l[4]), "+v"(v.l[5]), "+v"(v.l[6]), "+v"(v.l[7]), "+v"(v.l[8]))
// sum of v over the 32 lanes of the group (values < 2r, normalised limbs), valid in lane 0 of the group, < 64r
__device__ __forceinline__ fe group_sum32(fe v, int g) {
    ZK_FR_ROW_ADD_SHL(1); ZK_FR_ROW_ADD_SHL(2);      // four summands per limb: < 2^31
    fe_norm_u(v);
    ZK_FR_ROW_ADD_SHL(4); ZK_FR_ROW_ADD_SHL(8);      // lanes 0 and 16 of the group: the sums of their rows
    fe_norm_u(v);
    fe o;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)v.l[k], 16), hi = (u32)__builtin_amdgcn_readlane((int)v.l[k], 48);
        o.l[k] = g ? hi : lo;
    }
    return fe_add(v, o);
}
#undef ZK_FR_ROW_ADD_SHL
__device__ fe coop_poseidon_fr(fe x, u32* xs, u32 t) {
    const int l = threadIdx.x & 31, g = (threadIdx.x >> 5) & 1;
    const u32 lc = (u32)l < t ? l : 0;
    const Params P = g_prm[t - 2];
    x = fe_add(x, P.c[lc]);
    for (u32 r = 0; r < 3; ++r) {
        pow5(x); x = fe_add(x, P.c[(r + 1) * t + lc]);
        x = coop_matmul(P.m, x, xs, l, t);
    }
    pow5(x); x = fe_add(x, P.c[4 * t + lc]);
    x = coop_matmul(P.p, x, xs, l, t);
    const fe one = fe_one(), zero = fe_zero();
    for (u32 r = 0; r < P.n_rp; ++r) {
        const fe* __restrict__ S = P.s + (size_t)(2 * t - 1) * r;
        fe y = x;
        pow5(y); y = fe_add(y, P.c[5 * t + r]);                 // only lane 0's result is used: the new st[0]
        const fe st0 = group_bcast0(y, g);
        const fe prod = fe_pick((u32)l < t, fe_mul(S[lc], fe_pick(l == 0, st0, x)), zero);   // S[j] * st[j]; idle lanes add nothing
        const fe s0 = group_sum32(prod, g);                     // lane 0: < 34r
        // one multiplication serves both sides: lane 0 brings its sum back below 2r (x 1), lane k adds S[t + k - 1] * st[0]
        const fe m = fe_mul(fe_pick(l == 0, s0, S[t + (lc > 0 ? lc : 1) - 1]), fe_pick(l == 0, one, st0));
        x = fe_pick(l == 0, m, fe_add(x, m));
        if (r % PR_RENORM == PR_RENORM - 1) x = fe_pick(l == 0, x, fe_renorm(x));
    }
    x = fe_pick(l == 0, x, fe_renorm(x));
    for (u32 r = 0; r < 3; ++r) {
        pow5(x); x = fe_add(x, P.c[5 * t + P.n_rp + r * t + lc]);
        x = coop_matmul(P.m, x, xs, l, t);
    }
    pow5(x);
    return coop_matmul(P.m, x, xs, l, t);
}
__global__ __launch_bounds__(64) void bn128_level_coop_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    __shared__ u32 xs_all[2][17 * NR];
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    u32* xs = xs_all[g];
    const u64 i = (u64)blockIdx.x * 2 + g;
    const u64 ic = i < n_ops ? i : n_ops - 1;                   // an idle group shadows the last parent
    fe x = fe_zero();
    if (l >= 1 && l <= 16) x = load_raw(in + (ic * 16 + (l - 1)) * 4);
    x = coop_poseidon_fr(x, xs, 17);
    if (i < n_ops && l == FH_OUT_IDX) store_raw(x, out + 4 * i);
}
// LinearHash of a few, possibly very wide rows (the FRI trees of a large fold: final.starkStruct.*.json commits 2^7 rows of 3072
// words): one lane per row would run 64 sponge steps of 1.2 M instructions each on 128 lanes.  32 lanes per row instead, the
// sponge steps of linearhash_bn128.rs:105-131 in order, the digest handed from lane FH_OUT_IDX to lane 0 through LDS.
__global__ __launch_bounds__(64) void bn128_leaf_coop_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    __shared__ u32 xs_all[2][17 * NR];
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    u32* xs = xs_all[g];
    const u64 i = (u64)blockIdx.x * 2 + g;
    const u64 ic = i < height ? i : height - 1;                 // an idle group shadows the last row
    const u64* __restrict__ v = rows + ic * width;
    const u32 nb = (width - 1) / 3 + 1;
    fe digest = fe_zero();
    for (u32 b = 0; b < nb; b += 16) {
        const u32 sz = nb - b < 16 ? nb - b : 16;
        fe x = fe_zero();
        if (l == 0) x = digest;
        else if ((u32)l <= sz) {
            const u32 at = 3 * (b + l - 1), len = width - at < 3 ? width - at : 3;
            x = words_to_fe(v + at, len);
        }
        x = coop_poseidon_fr(x, xs, sz + 1);
        __syncthreads();
        if (l == FH_OUT_IDX) coop_put(xs, 0, 1, x);
        __syncthreads();
        digest = coop_get(xs, 0);
    }
    if (i < height && l == 0) store_raw(digest, digests + 4 * i);
}
// Poseidon::hash_ex for a handful of permutations (the transcript: one sponge step at a time): 32 lanes each
__global__ __launch_bounds__(64) void bn128_poseidon_coop_kernel(const u64* __restrict__ inp, u64 n, u32 n_in, const u64* __restrict__ init,
                                                                 u32 n_out, u64* __restrict__ out) {
    __shared__ u32 xs_all[2][17 * NR];
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    u32* xs = xs_all[g];
    const u64 i = (u64)blockIdx.x * 2 + g;
    const u64 ic = i < n ? i : n - 1;
    fe x = fe_zero();
    if (l == 0) x = load_raw(init);
    else if ((u32)l <= n_in) x = load_raw(inp + (ic * n_in + (l - 1)) * 4);
    x = coop_poseidon_fr(x, xs, n_in + 1);
    if (i < n && (u32)l < n_out) store_raw(x, out + (i * n_out + l) * 4);
}

struct DeviceTables { fe* all = nullptr; bool ready = false; };
DeviceTables g_tables[64];
const u32 NRP[16] = {FH_NRP};

void require_tables() {
    int dev; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64 && g_tables[dev].ready, FH_NAME " Poseidon constants not loaded (zk_" FH_NAME "_load_constants)");
}

}  // namespace

// file format: tools/gen_poseidon_bn128_constants.py
void FH_FN(load_constants)(const char* path) {
    int dev; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    if (g_tables[dev].ready) return;
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
    std::vector<unsigned char> canon;   // all 32-byte values back to back
    struct Off { size_t c, m, p, s; u32 t; } off[16];
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
    DevBuf d_canon; d_canon.reserve(canon.size());
    ZK_HIP(hipMemcpy(d_canon.p, canon.data(), canon.size(), hipMemcpyHostToDevice));
    fe* d_all = nullptr;
    ZK_HIP(hipMalloc((void**)&d_all, count * sizeof(fe)));
    hipLaunchKernelGGL(bn128_convert_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, (const u32*)d_canon.p, (u64)count, d_all);
    ZK_HIP(hipGetLastError());
    Params prm[16];
    for (int k = 0; k < 16; ++k) prm[k] = {off[k].t, NRP[k], d_all + off[k].c, d_all + off[k].m, d_all + off[k].p, d_all + off[k].s};
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
        hipLaunchKernelGGL(bn128_poseidon_coop_kernel, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, st, d_inp, n, n_in, d_init, n_out, d_out);
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
    const dim3 grid((unsigned)((height + 63) / 64)), blk(64);
    const u32 nb = width ? (width - 1) / 3 + 1 : 0;
    if (width > 4 && height <= 4096) {   // latency-bound: 32 lanes per row
        hipLaunchKernelGGL(bn128_leaf_coop_kernel, dim3((unsigned)((height + 1) / 2)), blk, 0, st, d_rows, width, height, d_digests);
        ZK_HIP(hipGetLastError());
        return;
    }
    static const int reg_max = getenv("ZK_FRHASH_REG_MAX") ? atoi(getenv("ZK_FRHASH_REG_MAX")) : 8;   // tuning knob: largest block count that takes the register kernels
    switch (width > 4 && (int)nb <= reg_max ? nb : 0) {   // one sponge step with the state in registers; wider rows (and width <= 4: no hash) take the generic kernel
        case 2: hipLaunchKernelGGL(bn128_leaf_reg_kernel<2>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        case 3: hipLaunchKernelGGL(bn128_leaf_reg_kernel<3>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        case 4: hipLaunchKernelGGL(bn128_leaf_reg_kernel<4>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        case 5: hipLaunchKernelGGL(bn128_leaf_reg_kernel<5>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        case 6: hipLaunchKernelGGL(bn128_leaf_reg_kernel<6>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        case 7: hipLaunchKernelGGL(bn128_leaf_reg_kernel<7>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        case 8: hipLaunchKernelGGL(bn128_leaf_reg_kernel<8>, grid, blk, 0, st, d_rows, width, height, d_digests); break;
        default: hipLaunchKernelGGL(bn128_leaf_kernel, grid, blk, 0, st, d_rows, width, height, d_digests);
    }
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
        static const u64 coop_upto = getenv("ZK_FR_LEVEL_COOP") ? strtoull(getenv("ZK_FR_LEVEL_COOP"), nullptr, 10) : 16384;
        if (next <= coop_upto)  // latency-bound: 32 lanes per parent
            hipLaunchKernelGGL(bn128_level_coop_kernel, dim3((unsigned)((next + 1) / 2)), dim3(64), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        else
            hipLaunchKernelGGL(bn128_level_kernel, dim3((unsigned)((next + 63) / 64)), dim3(64), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        ZK_HIP(hipGetLastError());
        n = next; next = (n - 1) / 16 + 1; p_in = p_out; p_out = p_in + next * 16;
    }
}

