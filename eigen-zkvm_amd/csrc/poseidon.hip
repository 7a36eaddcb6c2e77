// Poseidon-Goldilocks (t=12, x^7, 8 full + 22 partial rounds, "optimised" schedule), the
// LinearHash row digest and the binary Merkle tree built from them, for gfx950.
//
// Replaces starky/src/poseidon_opt.rs:80-200 (hash_inner), linearhash.rs:79-145 (hash/_hash),
// merklehash.rs:293-346 (merkelize), :79-134 (merklize_level), :47-61 (get_n_nodes).
//
// Mapping: one lane = one permutation, the 12-word state lives in 24 VGPRs from the first load to
// the digest store (the permutation is inlined once per kernel; a row's whole LinearHash is one
// loop around that single site).  The kernel is integer-ALU bound (2130 field multiplications per
// permutation, ~30 B of HBM traffic), so the work goes into the instruction count:
//   * the dense MDS has entries < 2^6, known at compile time: they become instruction immediates,
//     and each output word is two chains of 12 v_mad_u64_u32 over the 32-bit halves of the state
//     (no carries: 12 * 2^6 * 2^32 < 2^42) and ONE reduction instead of 12 modular multiplications;
//   * round 6: the products by full-width constants -- the pre-sparse matrix P and the two dense products of each lazy block
//     of partial rounds, 628 of a permutation's terms -- run on the MATRIX pipe as byte-limb i8 GEMMs over the wave's 64 states
//     (gl_mfma.hip.h): 3.9 k of 17.3 k vector instructions per permutation gone (SQ counters: 13.4 k), the S-boxes are what is left;
//   * the 64-bit round constants and the matrix-pipe tables sit in LDS (49 KB per workgroup of 256 lanes).
#include "zk_internal.h"
#include "poseidon_gl_constants.h"
#include "ntt_reg.hip.h"   // static_for
#include "acc6.hip.h"
#include "gl_mfma.hip.h"
#include <mutex>

namespace zk {

namespace {

// LDS image of the constant tables (u64 word offsets).  Constants that enter 12-term dot products (the
// pre-sparse matrix P, transposed, and the first 12 entries of every sparse row S) are stored split in
// three limbs of 22/22/20 bits, two words per constant: (l0 | l1 << 32, l2).
constexpr int T_C0 = 0;              // C[0..12): added before round 0
constexpr int T_FC = 12;             // [8][12]: constants added after the S-box of full round R (round 7: zeros)
constexpr int T_PC = T_FC + 96;      // [22] partial-round constants (+2 pad)
constexpr int T_PT = T_PC + 24;      // [12 i][12 j] split: P[j][i]
constexpr int T_SR = T_PT + 288;     // [22][12] split: S[23r + j], j < 12
constexpr int T_DD = T_SR + 528;     // [2 blocks][m (m - 1) / 2 + i] split: D[r][r0 + i], r = r0 + m (partial rounds, below)
constexpr int T_SCS = T_DD + 220;    // [2 blocks][11 k][11 m] split: S[23 (r0 + m) + 12 + k], the column entries regrouped by state word
constexpr int T_K0 = T_SCS + 484;    // [4]: C[8 + i]^7 + (round 0's constant of word 8 + i): what a zero capacity word is after the first S-box
constexpr int T_CD = T_K0 + 4;       // [2 blocks][11 m][16 l] split: what u_(r0 + m) weighs in lane l of a cooperative permutation (coop_partial_rounds)
constexpr int T_WORDS = T_CD + 704;  // 2360 words = 18.4 KB
constexpr int PR_B = 11;             // partial rounds per block
static_assert(T_PT % 2 == 0 && T_SR % 2 == 0 && T_DD % 2 == 0 && T_SCS % 2 == 0 && T_CD % 2 == 0, "split constants are read as 16-byte pairs");
__device__ u64 g_tab[T_WORDS];
#define ZK_POSEIDON_LDS __shared__ __attribute__((aligned(16))) u64 tab[T_WORDS]

// LDS image of the ONE-LANE kernels (g_mtab; round 6): the constants their vector instructions still read, then one matrix-pipe table
// (gl_mfma.hip.h) per dense product.
constexpr int TM_S0 = T_PT;                                        // [22] split: S_r[0]   (C0, FC, PC in front of it as in g_tab)
constexpr int TM_DD = TM_S0 + 44;                                  // as T_DD
constexpr int TM_K0 = TM_DD + 220;                                 // [12]: C[i]^7 + (round 0's constant of word i): what a ZERO input word is after the first S-box
constexpr int TM_HDR = TM_K0 + 12;                                 // 408 words
constexpr int MT_P = 0, MT_BS = 1, MT_BE = 3, MT_N = 5;            // tables: P, block starts 0 / 1, block ends 0 / 1
constexpr int TM_WORDS = TM_HDR + MT_N * pmfma::TAB_WORDS;          // 6288 words = 49 KB
static_assert(TM_HDR % 2 == 0 && TM_WORDS % 2 == 0 && TM_WORDS * 8 <= 65536, "one-lane tables: 16-byte aligned, inside 64 KB");
__host__ __device__ constexpr int tm_table(int k) { return TM_HDR + k * pmfma::TAB_WORDS; }
__device__ u64 g_mtab[TM_WORDS];
// Shape of the one-lane kernels: 256 lanes per workgroup, launch bounds of two waves per SIMD -- up to 256 registers keep a permutation's
// S-boxes, the eleven u's of a block and a product's accumulators out of the private segment (the kernels take 126-160, so three
// workgroups and their 3 x 49 KB of tables share a CU), and a wave that waits for the matrix pipe has partners filling the vector
// pipe.  Measured on 2^22 x 19 trees (profiles/r06/poseidon_mfma.md): 256 lanes 7.25 ms, 512 lanes 7.5, 384 lanes 9.2 (six-wave
// workgroups do not tile four SIMDs); bounds of three waves (168 registers) 8.2-9.5; resident grids (tables loaded once per CU)
// lose 5-10 % to one workgroup per chunk of rows -- workgroups that start together run their phases together.
constexpr int ONE_THREADS = 256;
#define ZK_ONE_BOUNDS __launch_bounds__(256, 2)
#define ZK_POSEIDON_LDS_ONE __shared__ __attribute__((aligned(16))) u64 tab[TM_WORDS]
__device__ __forceinline__ void load_tables_one(u64* __restrict__ tab) {
    for (int i = threadIdx.x; i < TM_WORDS / 2; i += blockDim.x) reinterpret_cast<ulonglong2*>(tab)[i] = reinterpret_cast<const ulonglong2*>(g_mtab)[i];
    __syncthreads();
}

__device__ __forceinline__ void load_tables(u64* __restrict__ tab) {
    for (int i = threadIdx.x; i < T_WORDS; i += blockDim.x) tab[i] = g_tab[i];
    __syncthreads();
}

// N terms: constants c[2 j], c[2 j + 1] (LDS), word j's halves from x(j).  The constants of four terms are requested together and one
// group ahead of the multiply-adds that use them (a read per term, waited for at once, left the LDS latency in the open 300 times
// per permutation).
template <int N, class X>
__device__ __forceinline__ void acc_dot(Acc6& A, const u64* __restrict__ c, X&& x) {
    constexpr int G = 4, NG = (N + G - 1) / G;
    ulonglong2 buf[2][G];
#pragma unroll
    for (int j = 0; j < G && j < N; ++j) buf[0][j] = reinterpret_cast<const ulonglong2*>(c)[j];
    static_for<0, NG>([&](auto GI) {
        constexpr int g = decltype(GI)::value;
        if constexpr (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < G && (g + 1) * G + j < N; ++j) buf[(g + 1) & 1][j] = reinterpret_cast<const ulonglong2*>(c)[(g + 1) * G + j];
        }
#pragma unroll
        for (int j = 0; j < G && g * G + j < N; ++j) {
            const ulonglong2 v = buf[g & 1][j];
            u32 x0, x1;
            x(g * G + j, x0, x1);
            const u32 l0 = (u32)v.x, l1 = (u32)(v.x >> 32), l2 = (u32)v.y;
            A.a00 += (u64)l0 * x0; A.a10 += (u64)l1 * x0; A.a20 += (u64)l2 * x0;
            A.a01 += (u64)l0 * x1; A.a11 += (u64)l1 * x1; A.a21 += (u64)l2 * x1;
        }
    });
}
__device__ __forceinline__ u64 dot12(const u64* __restrict__ c, const u32 (&x0)[12], const u32 (&x1)[12]) {
    Acc6 A; acc_zero(A);
    acc_dot<12>(A, c, [&](int j, u32& a, u32& b) { a = x0[j]; b = x1[j]; });
    return acc_finish(A);
}

// Inside a permutation every word is "nc" (gl.hip.h: some u64 congruent to the value); the dense MDS product and the
// batched dot products accept that and the last MDS of the permutation emits canonical words.
__device__ __forceinline__ u64 pow7(u64 x) {  // poseidon_opt.rs:68-74; any u64 in, nc out
    u64 x2 = gl::sqr_nc(x), x3 = gl::mul_nc(x2, x), x6 = gl::sqr_nc(x3);
    return gl::mul_nc(x6, x);
}
// x^7 + c for a canonical constant c: the addition rides on the multiply-adds of the last product
__device__ __forceinline__ u64 pow7_add(u64 x, u64 c) {
    u64 x2 = gl::sqr_nc(x), x3 = gl::mul_nc(x2, x), x6 = gl::sqr_nc(x3);
    return gl::mul_add_nc(x6, x, c);
}

// out[i] = sum_j M[j][i] * st[j]  (poseidon_opt.rs:111-119), M[j][i] < 2^6 folded to immediates
template <bool CANON, int N_OUT = 12>
__device__ __forceinline__ void mds_small(u64 (&st)[12]) {
    u32 lo32[12], hi32[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { lo32[j] = (u32)st[j]; hi32[j] = (u32)(st[j] >> 32); }
#pragma unroll
    for (int i = 0; i < N_OUT; ++i) {                                      // (the words past N_OUT keep their inputs: the caller does not read them)
        u64 lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const u32 m = (u32)ZK_POSEIDON_M[j * 12 + i];
            lo += (u64)lo32[j] * m;
            hi += (u64)hi32[j] * m;
        }
        // value = lo + hi * 2^32 with lo, hi < 2^42: words w0 = lo.lo, w1 = lo.hi + hi.lo (carry c), w2 = hi.hi + c < 2^11;
        // w2 * 2^64 = w2 * (2^32 - 1) rides on one multiply-add whose carry-out selects the fix-up (gl::mad_eps_nc)
        u32 c;
        const u32 w1 = __builtin_addc((u32)(lo >> 32), (u32)hi, 0u, &c);
        const u32 w2 = (u32)(hi >> 32) + c;
        const u64 r = gl::mad_eps_nc(w2, gl::mk64((u32)lo, w1));
        st[i] = (CANON && r >= GL_P) ? r - GL_P : r;                      // inside the permutation any representative does
    }
}

// st <- sum_j P[j][i] * st[j]  (poseidon_opt.rs:121-131) on the matrix pipe; every lane of the wave must be here
__device__ __forceinline__ void mat_full(const u64* __restrict__ mt /* LDS: table MT_P */, u64 (&st)[12]) {
    pmfma::BOps B;
    pmfma::make_b<12>(B, [&](int j) { return st[j]; });
    pmfma::product<3>(B, mt, [&](int o, u64 v) { st[o] = v; });
}

// The 22 partial rounds (poseidon_opt.rs:140-163) in two blocks of 11 without a reduction per state word and round.
// In round r only st[0] passes the S-box: u_r = st[0]^7 + c_r, st[0] <- S_r[0] u_r + sum_k S_r[k] st[k], st[k] += SC_r[k] u_r.
// The words k >= 1 are linear in the u's, so inside a block that starts at round r0 with words s_k
//     st[0] after round r  =  S_r[0] u_r + G[r] + sum_{r0 <= i < r} D[r][i] u_i,   G[r] = sum_k S_r[k] s_k,  D[r][i] = sum_k S_r[k] SC_i[k]
// (D precomputed on the host), and at the end of the block
//     s_k <- s_k + sum_i SC_i[k] u_i.
// Round 6: the block's two dense products -- the eleven G[r] before the first S-box, the eleven column updates after the last --
// are matrix-pipe products over the wave's 64 states (11 x 11 full-width constants each); what depends on the previous round -- the
// S-box chain, S_r[0] u_r and the D terms, batched dot products in 22/22/20-bit limbs with one recombination per round (acc6.hip.h) --
// stays on the vector pipe.  Blocks of 11 minimise the D terms + products (four blocks of 5-6 save 60 terms and cost four more products).
// tab = the TM_ image.
__device__ __forceinline__ void partial_rounds(u64 (&st)[12], const u64* __restrict__ tab) {
#pragma unroll 1
    for (int b = 0; b < 22 / PR_B; ++b) {
        const u64* __restrict__ S0 = tab + TM_S0 + 2 * PR_B * b;
        const u64* __restrict__ DD = tab + TM_DD + 2 * (PR_B * (PR_B - 1) / 2) * b;
        const u64* __restrict__ PC = tab + T_PC + PR_B * b;
        u32 u0[PR_B], u1[PR_B];
        u64 G[PR_B];
        {
            pmfma::BOps B;
            pmfma::make_b<11>(B, [&](int j) { return st[j + 1]; });
            pmfma::product<3>(B, tab + tm_table(MT_BS + b), [&](int o, u64 v) { if (o < PR_B) G[o] = v; });
        }
        u64 s0 = st[0];
        static_for<0, PR_B>([&](auto MI) {
            constexpr int m = decltype(MI)::value;
            const u64 u = pow7_add(s0, PC[m]);
            u0[m] = (u32)u; u1[m] = (u32)(u >> 32);
            Acc6 A;
            acc_word(A, G[m]);
            acc_mac(A, S0 + 2 * m, u0[m], u1[m]);
            if constexpr (m > 0) acc_dot<m>(A, DD + 2 * (m * (m - 1) / 2), [&](int i, u32& a, u32& c) { a = u0[i]; c = u1[i]; });
            s0 = acc_finish(A);
        });
        st[0] = s0;
        pmfma::BOps B;
        pmfma::make_b<11>(B, [&](int j) { return gl::mk64(u0[j], u1[j]); });
        pmfma::product_add<3>(B, tab + tm_table(MT_BE + b), [&](int o) { return st[o < PR_B ? o + 1 : 1]; },
                              [&](int o, u64 v) { if (o < PR_B) st[o + 1] = v; });
    }
}

// in-place permutation of st = in[8] || cap[4]   (poseidon_opt.rs:98-199); tab = the one-lane LDS image (TM_).  EVERY LANE OF THE
// WAVE must call it together (the matrix pipe works on the wave's 64 states at once): kernels keep idle lanes on a shadow input.
// Two things every LinearHash and every tree node allow (round 3):
//   * zero_cap (wave-uniform): the capacity words are zero -- the first block of a sponge and every node of a tree.  After the
//     first constants they are C[8..12) whatever the input, so their first S-boxes are four table words (TM_K0), not 16 products;
//   * FULL_OUT = false: only st[0..4) is read afterwards (a digest, or the capacity of the next block): the last MDS computes
//     four of its twelve outputs.  st[4..12) are then NOT the permutation's words.
//   * n_in (wave-uniform, round 6): only st[0..n_in) hold input, the rest of the rate is the sponge's zero padding (the last block of a
//     batch whose length is not a multiple of 8, the second half of the digests' sponge): their first S-boxes are table words too.
//     A row of 36 columns absorbs four such blocks of ONE word each.
template <bool FULL_OUT = false>
__device__ __forceinline__ void poseidon_perm(u64 (&st)[12], const u64* __restrict__ tab, bool zero_cap, u32 n_in = 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if ((u32)i < n_in) st[i] = pow7_add(gl::add_nc(st[i], tab[T_C0 + i]), tab[T_FC + i]);
        else st[i] = tab[TM_K0 + i];
    }
    if (zero_cap) {
#pragma unroll
        for (int i = 8; i < 12; ++i) st[i] = tab[TM_K0 + i];
    } else {
#pragma unroll
        for (int i = 8; i < 12; ++i) st[i] = pow7_add(gl::add_nc(st[i], tab[T_C0 + i]), tab[T_FC + i]);
    }
    mds_small<false>(st);
#pragma unroll 1
    for (int R = 1; R < 7; ++R) {
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = pow7_add(st[i], tab[T_FC + R * 12 + i]);
        if (R != 3) { mds_small<false>(st); continue; }
        mat_full(tab + tm_table(MT_P), st);
        partial_rounds(st, tab);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = pow7(st[i]);
    mds_small<true, FULL_OUT ? 12 : 4>(st);
}


// ---- cooperative permutation: 16 lanes per permutation, lane l < 12 owns state word l ---------------------
// A single-lane permutation is a 33 k-instruction dependent chain (~100 us).  The small levels of a Merkle
// tree, the transcript and every other place where fewer than a few thousand permutations are in flight are
// bound by that latency, not by throughput; spreading one state over 12 lanes cuts the chain to ~1/5.
// Words travel by ds_bpermute (__shfl, width 16); constants as in the one-lane version.
static_assert(ZK_POSEIDON_M[0] == ZK_POSEIDON_M[13] + 8, "MDS = circulant + 8 * E00");
__host__ __device__ constexpr bool mds_is_circulant() {
    for (int j = 0; j < 12; ++j)
        for (int i = 0; i < 12; ++i) {
            const u64 want = ZK_POSEIDON_M[((j - i + 12) % 12) * 12] - ((j - i + 12) % 12 == 0 ? 8 : 0) + (i == 0 && j == 0 ? 8 : 0);
            if (ZK_POSEIDON_M[j * 12 + i] != want) return false;
        }
    return true;
}
static_assert(mds_is_circulant(), "out[i] = sum_k circ[k] * st[(i + k) % 12] (+ 8 st[0] for i = 0)");
__device__ __forceinline__ u64 shfl64(u64 v, int src) {
    const u32 lo = (u32)__shfl((int)(u32)v, src, 16), hi = (u32)__shfl((int)(u32)(v >> 32), src, 16);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 dpp_opaque(u32 v) { asm volatile("" : "+v"(v)); return v; }   // keeps the DPP moves where they are written (section 3.9)
// lane SRC of every 16-lane row to the whole row: one DPP move per half (row_newbcast, gfx90a and later) -- a VALU move, where
// __shfl(.., SRC, 16) is a ds_bpermute round trip through the LDS crossbar (~120 cycles on the critical path of a lone wave)
template <int SRC>
__device__ __forceinline__ u64 bcast64(u64 v) {
    static_assert(SRC >= 0 && SRC < 16, "a lane of the row");
    const u32 v0 = dpp_opaque((u32)v), v1 = dpp_opaque((u32)(v >> 32));
    const u32 z0 = dpp_opaque((u32)__builtin_amdgcn_update_dpp(0, (int)v0, 0x150 + SRC, 0xF, 0xF, false));
    const u32 z1 = dpp_opaque((u32)__builtin_amdgcn_update_dpp(0, (int)v1, 0x150 + SRC, 0xF, 0xF, false));
    return gl::mk64(z0, z1);
}
__device__ __forceinline__ u64 coop_mds(u64 x, int l) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    const u32 c0 = (u32)ZK_POSEIDON_M[13] + (l == 0 ? 8u : 0u);
    u64 alo = (u64)lo * c0, ahi = (u64)hi * c0;
#pragma unroll
    for (int k = 1; k < 12; ++k) {
        const int src = l + k >= 12 ? l + k - 12 : l + k;
        const u32 m = (u32)ZK_POSEIDON_M[k * 12];                    // circ[k], an immediate
        alo += (u64)(u32)__shfl((int)lo, src, 16) * m;
        ahi += (u64)(u32)__shfl((int)hi, src, 16) * m;
    }
    u32 c;                                              // as in mds_small
    const u32 w1 = __builtin_addc((u32)(alo >> 32), (u32)ahi, 0u, &c);
    const u32 w2 = (u32)(ahi >> 32) + c;
    const u64 r = gl::mad_eps_nc(w2, gl::mk64((u32)alo, w1));
    return r >= GL_P ? r - GL_P : r;
}
__device__ __forceinline__ void coop_gather(u64 x, u32 (&x0)[12], u32 (&x1)[12]) {
    static_for<0, 12>([&](auto JI) {
        constexpr int j = decltype(JI)::value;
        const u64 b = bcast64<j>(x);
        x0[j] = (u32)b; x1[j] = (u32)(b >> 32);
    });
}
// x^7 + c where the two lanes of a pair hold the same x: the even lane takes x^3 = x^2 x, the odd one x^4 = x^2 x^2, each fetches the
// other's (one DPP move per half) and both end with x^3 x^4 + c -- three dependent products where pow7_add has four.  The words of
// a product are the digits of the exact integer, so both lanes hold the same bits.
__device__ __forceinline__ u64 pow7_add_pair(u64 x, u64 c, bool odd) {
    const u64 x2 = gl::mul_nc(x, x);
    const u64 y = gl::mul_nc(x2, odd ? x2 : x);
    const u32 y0 = dpp_opaque((u32)y), y1 = dpp_opaque((u32)(y >> 32));
    const u32 z0 = dpp_opaque((u32)__builtin_amdgcn_update_dpp((int)y0, (int)y0, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    const u32 z1 = dpp_opaque((u32)__builtin_amdgcn_update_dpp((int)y1, (int)y1, 0xB1, 0xF, 0xF, false));
    return gl::mul_add_nc(y, gl::mk64(z0, z1), c);
}
// The 22 partial rounds of a cooperative permutation, in the two lazy blocks of partial_rounds() with the rounds of a block spread
// over the lanes: lane m < 11 keeps the accumulator of st[0] after round r0 + m,
//     sum_k S_r[k] s_k  (on entry, s = the block's first state)  +  sum_{i < m} D[r][r0 + i] u_i  +  S_r[0] u_m,
// every lane computes every S-box (the same s0 everywhere: nothing to broadcast before it), adds its own multiple of u_m (T_CD)
// and recombines; lane m's result is the next s0 and reaches the row through the LDS crossbar.  At the block's end lane k >= 1
// adds sum_m SC_m[k] u_m to its own word.  Per round: one S-box of three products, six multiply-adds, one recombination and
// one exchange -- the round-by-round form had the S-box of four, two broadcasts, a 128-bit row reduction and a product per lane.
__device__ __forceinline__ u64 coop_partial_rounds(u64 x, const u64* __restrict__ tab, int l) {
    const int lm = l < PR_B ? l : PR_B - 1;            // lanes 11..15: a copy of lane 10's accumulator, never read
    const int lk = l >= 1 && l < 12 ? l : 1;           // the state word this lane updates (lane 0 takes s0, lanes 12..15 carry garbage)
    const bool odd = l & 1;
#pragma unroll 1
    for (int b = 0; b < 22 / PR_B; ++b) {
        u32 x0[12], x1[12];
        coop_gather(x, x0, x1);
        Acc6 A; acc_zero(A);
        acc_dot<11>(A, tab + T_SR + 24 * (PR_B * b + lm) + 2, [&](int k, u32& a, u32& c) { a = x0[k + 1]; c = x1[k + 1]; });
        u64 s0 = gl::mk64(x0[0], x1[0]);
        const u64* __restrict__ CD = tab + T_CD + 2 * (16 * PR_B * b + l);
        const u64* __restrict__ PC = tab + T_PC + PR_B * b;
        u32 u0[PR_B], u1[PR_B];
        // a round's two constants are requested a round ahead, in front of the recombination and the exchange that end the previous round (left to
        // itself the compiler reads them where they are used and the LDS latency stands in the open twice per round)
        ulonglong2 cd = *reinterpret_cast<const ulonglong2*>(CD);
        u64 pc = PC[0];
        static_for<0, PR_B>([&](auto MI) {
            constexpr int m = decltype(MI)::value;
            const u64 u = pow7_add_pair(s0, pc, odd);
            u0[m] = (u32)u; u1[m] = (u32)(u >> 32);
            acc_mac(A, cd, u0[m], u1[m]);
            if constexpr (m + 1 < PR_B) { cd = *reinterpret_cast<const ulonglong2*>(CD + 32 * (m + 1)); pc = PC[m + 1]; }
            __builtin_amdgcn_sched_barrier(0);
            s0 = bcast64<m>(acc_finish(A));                                  // lane m's accumulator is the next round's s0 in every lane of the row
            __builtin_amdgcn_sched_barrier(0);
        });
        Acc6 E; acc_word(E, x);
        acc_dot<PR_B>(E, tab + T_SCS + 2 * (11 * PR_B * b + PR_B * (lk - 1)), [&](int m, u32& a, u32& c) { a = u0[m]; c = u1[m]; });
        const u64 e = acc_finish(E);
        x = l == 0 ? s0 : e;
    }
    return x;
}

// x = this lane's state word (lanes 12..15 of a group carry garbage and only serve the shuffles)
__device__ __forceinline__ u64 coop_perm(u64 x, const u64* __restrict__ tab) {
    const int l = threadIdx.x & 15, lc = l < 12 ? l : 11;
    x = gl::add_nc(x, tab[T_C0 + lc]);
#pragma unroll 1
    for (int R = 0; R < 8; ++R) {
        x = pow7_add(x, tab[T_FC + R * 12 + lc]);
        if (R != 3) { x = coop_mds(x, l); continue; }
        {
            u32 x0[12], x1[12];
            coop_gather(x, x0, x1);
            x = dot12(tab + T_PT + 24 * lc, x0, x1);
        }
        x = coop_partial_rounds(x, tab, l);
    }
    return x;
}

// LinearHash of one row (linearhash.rs:79-145), one loop around the single inlined permutation.
// hash(): bs = max(8, ceil(w/4)) words per batch; each batch is digested by _hash (rate-8 sponge,
// capacity carried, tail zero-padded; a batch of <= 4 words is its own zero-padded digest -- only
// the last batch can be that short); more than one batch digest -> _hash over the digests.
// All control flow depends on w only, i.e. is wave-uniform.
__device__ __forceinline__ void linearhash_row(const u64* __restrict__ row, u32 w, u64 (&out)[4], const u64* __restrict__ tab) {
    if (w <= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = (u32)i < w ? row[i] : 0;
        return;
    }
    u32 bs = (w + 3) / 4; if (bs < 8) bs = 8;
    const u32 hsz = (w + bs - 1) / bs;  // 1..4 batch digests
    u64 h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0;
    u64 st[12];
#pragma unroll
    for (int i = 8; i < 12; ++i) st[i] = 0;
    u32 b = 0, off = 0;
    bool final_sponge = false, second = false, cz = true;   // cz: the capacity words are zero (first block of a sponge)
    for (;;) {
        u32 n_in;                                              // how many words of the rate hold input (the rest is the sponge's zero padding)
        if (!final_sponge) {
            const u32 len = (w - b * bs < bs) ? w - b * bs : bs;
            const u64* __restrict__ v = row + (u64)b * bs + off;
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = (off + i < len) ? v[i] : 0;
            n_in = len - off < 8 ? len - off : 8;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = second ? h[8 + i] : h[i];
            n_in = second ? 4 * hsz - 8 : (4 * hsz < 8 ? 4 * hsz : 8);
        }
        poseidon_perm(st, tab, cz, n_in);
        if (!final_sponge) {
            const u32 len = (w - b * bs < bs) ? w - b * bs : bs;
            off += 8;
            if (off < len) {
#pragma unroll
                for (int i = 0; i < 4; ++i) st[8 + i] = st[i];
                cz = false;
                continue;
            }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                if ((u32)bb == b) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[4 * bb + i] = st[i];
                }
            ++b; off = 0;
            if (b < hsz && w - b * bs <= 4) {  // short last batch: identity padding, no permutation
                const u32 len2 = w - b * bs;
                const u64* __restrict__ v = row + (u64)b * bs;
#pragma unroll
                for (int bb = 1; bb < 4; ++bb)
                    if ((u32)bb == b) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) h[4 * bb + i] = (u32)i < len2 ? v[i] : 0;
                    }
                ++b;
            }
#pragma unroll
            for (int i = 8; i < 12; ++i) st[i] = 0;
            cz = true;
            if (b < hsz) continue;
            if (hsz == 1) break;      // st[0..4) is the digest
            final_sponge = true;
        } else {
            if (second || hsz <= 2) break;
            second = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) st[8 + i] = st[i];
            cz = false;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = st[i];
}

// The same digest in two launches for trees of middling height: one lane per (row, batch) hashes its batch, then one lane per
// row sponges the batch digests.  A row of 37 words is 4 batches of 2 permutations and a final sponge of 2: four
// permutations deep instead of ten -- what counts while there are too few rows to fill the chip (2^15-row proof 5.47 -> 5.36 ms,
// 2^18-row proof 11.64 -> 11.41 ms; from 2^19 rows on the one-launch kernel is the faster one).
// One lane per HASHED batch: a last batch of <= 4 words is its own digest (linearhash.rs:121-126) and is written by the lane of the batch
// before it; every lane of a wave runs the same number of permutations (the matrix pipe's condition): a shorter last batch idles a trip.
__global__ ZK_ONE_BOUNDS void linearhash_batch_kernel(const u64* __restrict__ rows, u32 w, u64 height, u32 bs, u32 hsz, u32 n_hashed,
                                                      u64* __restrict__ h /* [height][hsz][4] */) {
    ZK_POSEIDON_LDS_ONE;
    load_tables_one(tab);
    const u64 t_ = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t_ < height * n_hashed;
    const u64 t = live ? t_ : height * n_hashed - 1;                   // idle lanes shadow the last batch
    const u64 r = t / n_hashed;
    const u32 b = (u32)(t - r * n_hashed);
    const u32 len = (w - b * bs < bs) ? w - b * bs : bs;               // > 4
    const u64* __restrict__ v = rows + r * w + (u64)b * bs;
    u64 st[12], keep[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 8; i < 12; ++i) st[i] = 0;
    for (u32 off = 0; off < bs; off += 8) {
        const bool act = off < len;
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = (off + i < len) ? v[off + i] : 0;
        poseidon_perm(st, tab, off == 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (act) keep[i] = st[i]; st[8 + i] = st[i]; }   // the capacity carries the digest so far
    }
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[(r * hsz + b) * 4 + i] = keep[i];
    if (b + 1 == n_hashed && n_hashed < hsz) {
        const u32 len2 = w - hsz * bs + bs;                             // the short last batch
#pragma unroll
        for (int i = 0; i < 4; ++i) h[(r * hsz + hsz - 1) * 4 + i] = (u32)i < len2 ? v[bs + i] : 0;
    }
}
__global__ ZK_ONE_BOUNDS void linearhash_final_kernel(const u64* __restrict__ h, u32 hsz, u64 height, u64* __restrict__ digests) {
    ZK_POSEIDON_LDS_ONE;
    load_tables_one(tab);
    const u64 r_ = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 r = r_ < height ? r_ : height - 1;                       // idle lanes shadow the last row
    const u64* __restrict__ v = h + r * hsz * 4;
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = (u32)i < 4 * hsz ? v[i] : 0;
#pragma unroll
    for (int i = 8; i < 12; ++i) st[i] = 0;
    poseidon_perm(st, tab, true, 4 * hsz < 8 ? 4 * hsz : 8);
    if (hsz > 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st[8 + i] = st[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = 8 + (u32)i < 4 * hsz ? v[8 + i] : 0;
        poseidon_perm(st, tab, false, 4 * hsz - 8);
    }
    if (r_ >= height) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) digests[4 * r + i] = st[i];
}
__global__ ZK_ONE_BOUNDS void linearhash_rows_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    ZK_POSEIDON_LDS_ONE;
    load_tables_one(tab);
    const u64 r_ = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 r = r_ < height ? r_ : height - 1;                       // idle lanes shadow the last row: the matrix pipe wants whole waves
    u64 d[4];
    linearhash_row(rows + r * width, width, d, tab);
    if (r_ >= height) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) digests[4 * r + i] = d[i];
}

// The same LinearHash with 16 lanes per row (coop_perm): lane l < 12 owns state word l, lanes 0..3 keep the batch digests.
// For trees of few rows -- the FRI steps' trees: 2^3 ... 2^13 rows of up to 3 * 2^6 words, i.e. a sponge of up to 26
// permutations per row -- the one-lane kernel is a single dependent chain of ~33 k instructions per permutation on a handful
// of waves; spreading the state cuts the chain to a fifth.  Same control flow as linearhash_row (it depends on w only).
__device__ __forceinline__ u64 coop_linearhash_row(const u64* __restrict__ row, u32 w, int l, const u64* __restrict__ tab) {
    if (w <= 4) return (u32)l < w ? row[l] : 0;
    u32 bs = (w + 3) / 4; if (bs < 8) bs = 8;
    const u32 hsz = (w + bs - 1) / bs;
    u64 hq[4] = {0, 0, 0, 0};                 // lane l < 4: word l of batch digest 0..3
    u64 x = 0, cap = 0;                       // cap: lanes 8..11 carry the capacity into the next permutation
    u32 b = 0, off = 0;
    bool final_sponge = false, second = false;
    for (;;) {
        if (!final_sponge) {
            const u32 len = (w - b * bs < bs) ? w - b * bs : bs;
            const u64* __restrict__ v = row + (u64)b * bs + off;
            x = l < 8 ? (off + l < len ? v[l] : 0) : cap;
        } else {
            const u64 a = shfl64(second ? hq[2] : hq[0], l & 3), c = shfl64(second ? hq[3] : hq[1], l & 3);
            x = l < 4 ? a : l < 8 ? c : cap;
        }
        x = coop_perm(x, tab);
        if (!final_sponge) {
            const u32 len = (w - b * bs < bs) ? w - b * bs : bs;
            off += 8;
            if (off < len) { cap = shfl64(x, (l + 8) & 15); continue; }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) if ((u32)bb == b) hq[bb] = x;
            ++b; off = 0;
            if (b < hsz && w - b * bs <= 4) {   // short last batch: identity padding, no permutation
                const u32 len2 = w - b * bs;
                const u64 v = (u32)l < len2 ? row[(u64)b * bs + l] : 0;
#pragma unroll
                for (int bb = 1; bb < 4; ++bb) if ((u32)bb == b) hq[bb] = v;
                ++b;
            }
            cap = 0;
            if (b < hsz) continue;
            if (hsz == 1) break;
            final_sponge = true;
        } else {
            if (second || hsz <= 2) break;
            second = true;
            cap = shfl64(x, (l + 8) & 15);
        }
    }
    return x;
}
__global__ __launch_bounds__(256) void linearhash_rows_coop_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    ZK_POSEIDON_LDS;
    load_tables(tab);
    const int l = threadIdx.x & 15;
    const u64 r = (u64)blockIdx.x * 16 + (threadIdx.x >> 4);
    const u64 rc = r < height ? r : height - 1;                  // idle groups shadow the last row (no divergence)
    const u64 x = coop_linearhash_row(rows + rc * width, width, l, tab);
    if (r < height && l < 4) digests[4 * r + l] = x;
}

// rows of at most four words are their own zero-padded digest (linearhash.rs:85-91): no permutation, no tables
__global__ __launch_bounds__(256) void linearhash_pad_kernel(const u64* __restrict__ rows, u32 w, u64 height, u64* __restrict__ digests) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * height) return;
    const u32 i = (u32)(t & 3);
    digests[t] = i < w ? rows[(t >> 2) * w + i] : 0;
}

// One WAVE per row, for the trees of the fewest rows (FRI steps: 2^3 ... 2^11 rows of 48 ... 192 words).  The up to four batches of a
// row's LinearHash are independent sponges (linearhash.rs:79-145): group g of 16 lanes digests batch g, then every group runs the
// sponge over the digests -- a row of 96 words is 3 + 2 permutations deep where the 16-lane form above walks its 14 one after
// the other.  Four times the lanes for the same work: only where the chip is far from full.
__global__ __launch_bounds__(256) void linearhash_rows_wave_kernel(const u64* __restrict__ rows, u32 w, u64 height, u64* __restrict__ digests) {
    ZK_POSEIDON_LDS;
    load_tables(tab);
    const int l = threadIdx.x & 15;
    const u32 g = (threadIdx.x >> 4) & 3;
    const u64 r = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u64* __restrict__ row = rows + (r < height ? r : height - 1) * w;   // idle waves shadow the last row
    const u32 bs = std::max<u32>(8, (w + 3) / 4), hsz = (w + bs - 1) / bs;   // 2 <= hsz <= 4 here
    const u32 gb = g < hsz ? g : hsz - 1;                                    // idle groups shadow the last batch
    const u32 len = std::min<u32>(bs, w - gb * bs);
    const u64* __restrict__ v = row + (u64)gb * bs;
    const bool hashed = len > 4;                                             // a last batch of <= 4 words is its own digest
    u64 d = !hashed && (u32)l < len ? v[l] : 0, cap = 0;
    for (u32 off = 0; off < bs; off += 8) {                                  // the same trip count in every group: the short batch idles
        const u64 y = coop_perm(l < 8 ? (off + l < len ? v[off + l] : 0) : cap, tab);
        if (hashed && off < len) d = y;
        cap = shfl64(d, (l + 8) & 15);
    }
    if (g >= hsz) d = 0;
    const auto digest_word = [&](u32 first) {                               // lanes 0..7 <- words of digests first, first + 1
        const int src = 16 * (int)(first + ((l >> 2) & 1)) + (l & 3);
        const u32 lo = (u32)__shfl((int)(u32)d, src, 64), hi = (u32)__shfl((int)(u32)(d >> 32), src, 64);
        return ((u64)hi << 32) | lo;
    };
    u64 x = coop_perm(l < 8 ? digest_word(0) : 0, tab);
    if (hsz > 2) {
        cap = shfl64(x, (l + 8) & 15);
        const u64 in = digest_word(2);
        x = coop_perm(l < 8 ? in : cap, tab);
    }
    if (r < height && g == 0 && l < 4) digests[4 * r + l] = x;
}

// merklehash.rs:110-134 do_merklize_level: parent i = Poseidon(node[2i] || node[2i+1], cap 0)
__global__ ZK_ONE_BOUNDS void merkle_level_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    ZK_POSEIDON_LDS_ONE;
    load_tables_one(tab);
    const u64 i_ = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 i = i_ < n_ops ? i_ : n_ops - 1;                         // idle lanes shadow the last parent
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = in[8 * i + k];
#pragma unroll
    for (int k = 8; k < 12; ++k) st[k] = 0;
    poseidon_perm(st, tab, true);
    if (i_ >= n_ops) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * i + k] = st[k];
}

// the same for small levels: 16 lanes per parent (coop_perm), 16 parents per block
__global__ __launch_bounds__(256) void merkle_level_coop_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    ZK_POSEIDON_LDS;
    load_tables(tab);
    const int l = threadIdx.x & 15;
    const u64 i = (u64)blockIdx.x * 16 + (threadIdx.x >> 4);
    const u64 ic = i < n_ops ? i : n_ops - 1;                  // idle groups shadow the last parent (no divergence)
    u64 x = l < 8 ? in[8 * ic + l] : 0;
    x = coop_perm(x, tab);
    if (i < n_ops && l < 4) out[4 * i + l] = x;
}

// The top of a tree in ONE launch: from n <= 64 children to the root, one group of 16 lanes per parent, a block barrier
// between the levels.  Each of these levels is one permutation deep whatever runs it; as launches of their own they cost
// a dependent-launch gap each (six per tree, and a small proof builds seven trees or more): 2^15-row proof 6.79 -> 6.69 ms.
__global__ __launch_bounds__(1024) void merkle_top_coop_kernel(u64* nodes, u64 n, u64 p_in) {
    ZK_POSEIDON_LDS;
    load_tables(tab);
    const int l = threadIdx.x & 15;
    const u64 g = threadIdx.x >> 4;
    while (n > 1) {                                                // merklehash.rs:331-343
        const u64 next = (n - 1) / 2 + 1, p_out = p_in + 2 * next;
        if ((g & ~(u64)3) < next) {                                // a wave whose four groups are all idle only waits
            const u64 gc = g < next ? g : next - 1;                // idle groups of a busy wave shadow the last parent (no divergence)
            u64 x = l < 8 ? nodes[4 * p_in + 8 * gc + l] : 0;
            x = coop_perm(x, tab);
            if (g < next && l < 4) nodes[4 * p_out + 4 * g + l] = x;
        }
        __syncthreads();                                           // the level is in memory before the block reads it back
        n = next; p_in = p_out;
    }
}

// The verifier's side of a tree: calculate_root_from_group_proof (merklehash.rs:393-428 -> merkle_calculate_root_from_proof): from a
// leaf digest up its path, (value, sibling) ordered by the index bit of the level.  A proof opens a handful of paths, each a chain of
// dependent permutations: 16 lanes per path (coop_perm), four paths per wave; the paths of one launch may have different depths.
__global__ __launch_bounds__(64) void merkle_root_from_path_kernel(const u64* __restrict__ leaves /* [n][4] */, const u64* __restrict__ paths /* [n][max_depth][4] */,
                                                                   const u32* __restrict__ depth, const u64* __restrict__ idx, u32 n, u32 max_depth,
                                                                   u64* __restrict__ roots /* [n][4] */) {
    ZK_POSEIDON_LDS;
    load_tables(tab);
    __syncthreads();
    const int l = threadIdx.x & 15;
    const u32 g = blockIdx.x * 4 + (threadIdx.x >> 4);
    const u32 gc = g < n ? g : n - 1;                               // idle groups shadow the last path (no divergence inside a wave)
    u32 wave_depth = 0;                                             // every group of the wave runs the longest of its four chains
    for (u32 k = 0; k < 4; ++k) { const u32 gk = blockIdx.x * 4 + k; const u32 d = depth[gk < n ? gk : n - 1]; wave_depth = d > wave_depth ? d : wave_depth; }
    const u32 my_depth = depth[gc];
    u64 i = idx[gc];
    u64 cur = l < 4 ? leaves[4 * (u64)gc + l] : 0;                 // lanes 0..3 hold the running digest
    for (u32 lv = 0; lv < wave_depth; ++lv) {
        const bool live = lv < my_depth;
        const u64 sib = l < 4 && live ? paths[((u64)gc * max_depth + lv) * 4 + l] : 0;
        const bool right = (i & 1) != 0;                            // this node is the right child: (sibling, value)
        // lane j < 4 holds value[j] and sibling[j]; the permutation's input word k is value/sibling word k & 3
        const u64 v = shfl64(cur, l & 3), sb = shfl64(sib, l & 3);   // (a shuffle inside the 16-lane group)
        u64 x = l < 4 ? (right ? sb : v) : l < 8 ? (right ? v : sb) : 0;
        x = coop_perm(x, tab);
        if (live) { cur = x; i >>= 1; }
    }
    if (g < n && l < 4) roots[4 * (u64)g + l] = cur;
}

// A tree over zero-width rows (tree2 / tree3 of a PIL without plookups or grand products,
// stark_gen.rs:311,359) has all-zero leaves, so every node of a level holds the same digest:
// one permutation per level instead of one per node.
__global__ __launch_bounds__(64) void zero_tree_chain_kernel(u32 levels, u64* __restrict__ h /* [levels + 1][4] */) {
    ZK_POSEIDON_LDS_ONE;
    load_tables_one(tab);
    const bool w = threadIdx.x == 0 && blockIdx.x == 0;              // (all 64 lanes run the same permutation: the matrix pipe wants the whole wave)
    u64 cur[4] = {0, 0, 0, 0};
    if (w) for (int k = 0; k < 4; ++k) h[k] = 0;
    for (u32 l = 0; l < levels; ++l) {
        u64 st[12];
        for (int k = 0; k < 4; ++k) { st[k] = cur[k]; st[4 + k] = cur[k]; st[8 + k] = 0; }
        poseidon_perm(st, tab, true);
        for (int k = 0; k < 4; ++k) { cur[k] = st[k]; if (w) h[4 * (l + 1) + k] = st[k]; }
    }
}
__global__ void fill_digest_kernel(u64* __restrict__ nodes, u64 n, const u64* __restrict__ h) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * n) nodes[i] = h[i & 3];
}

__global__ __launch_bounds__(64) void poseidon_one_kernel(const u64* in8, const u64* cap4, u64* out, int n_out) {
    ZK_POSEIDON_LDS_ONE;
    load_tables_one(tab);
    u64 st[12];                                                       // (all 64 lanes run the same permutation, lane 0 writes)
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = in8[k] >= GL_P ? in8[k] - GL_P : in8[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) st[8 + k] = cap4[k] >= GL_P ? cap4[k] - GL_P : cap4[k];
    poseidon_perm<true>(st, tab, false);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < n_out) out[k] = st[k];
}

// ---- TranscriptGL (transcript.rs:8-103), device resident ---------------------------------------
// The Fiat-Shamir sponge lives in HBM next to the data it absorbs (roots, evals, final
// polynomial) and next to the kernels that consume its challenges, so a proof is one stream of
// launches with no host round trip until the query indices are needed.  One lane does the work
// (a sponge is inherently serial); state layout = TranscriptState below.
struct TranscriptState { u64 state[4]; u64 pending[8]; u64 out[12]; u32 n_pending, out_pos, n_out, _pad; };

// One wave runs the sponge: lanes 0..15 carry the permutation (coop_perm), lane 0 keeps the counters.  The
// state is staged in LDS for the duration of a kernel so that all lanes see lane 0's bookkeeping.
__device__ __forceinline__ void tr_load(TranscriptState* ts, const TranscriptState* t) {
    const u32* g = reinterpret_cast<const u32*>(t); u32* s = reinterpret_cast<u32*>(ts);
    for (u32 k = threadIdx.x; k < sizeof(TranscriptState) / 4; k += blockDim.x) s[k] = g[k];
    __syncthreads();
}
__device__ __forceinline__ void tr_store(TranscriptState* t, const TranscriptState* ts) {
    __syncthreads();
    u32* g = reinterpret_cast<u32*>(t); const u32* s = reinterpret_cast<const u32*>(ts);
    for (u32 k = threadIdx.x; k < sizeof(TranscriptState) / 4; k += blockDim.x) g[k] = s[k];
}
// transcript.rs:15-24; called by every lane of the wave (uniform control flow)
__device__ __forceinline__ void tr_update(TranscriptState* ts, const u64* __restrict__ tab) {
    const int l = threadIdx.x & 15;
    u64 x = 0;
    if (l < 8) x = (u32)l < ts->n_pending ? ts->pending[l] : 0;
    else if (l < 12) x = ts->state[l - 8];
    x = coop_perm(x, tab);
    __syncthreads();
    if (threadIdx.x < 12) ts->out[threadIdx.x] = x;
    if (threadIdx.x < 4) ts->state[threadIdx.x] = x;
    if (threadIdx.x == 0) { ts->n_pending = 0; ts->out_pos = 0; ts->n_out = 12; }
    __syncthreads();
}
__global__ void tr_init_kernel(TranscriptState* t) {
    if (threadIdx.x | blockIdx.x) return;
    for (int i = 0; i < 4; ++i) t->state[i] = 0;
    t->n_pending = 0; t->out_pos = 0; t->n_out = 0;
}
__global__ __launch_bounds__(64) void tr_put_kernel(TranscriptState* t, const u64* __restrict__ src, u64 n) {  // transcript.rs:25-33,64-71
    ZK_POSEIDON_LDS;
    __shared__ TranscriptState ts;
    load_tables(tab);
    tr_load(&ts, t);
    for (u64 i = 0; i < n; ++i) {
        if (threadIdx.x == 0) { ts.n_out = 0; ts.out_pos = 0; ts.pending[ts.n_pending++] = src[i]; }
        __syncthreads();
        if (ts.n_pending == 8) tr_update(&ts, tab);
    }
    tr_store(t, &ts);
}
// squeeze n_words (transcript.rs:54-62 get_fields1, repeated); get_field = 3 words.  `bits` != 0
// turns the squeezed words into n query indices of `bits` bits (get_permutations, :73-102).
__global__ __launch_bounds__(64) void tr_get_kernel(TranscriptState* t, u64* __restrict__ dst, u32 n, u32 bits) {
    ZK_POSEIDON_LDS;
    __shared__ TranscriptState ts;
    load_tables(tab);
    tr_load(&ts, t);
    if (bits == 0) {
        for (u32 i = 0; i < n; ++i) {
            if (ts.out_pos >= ts.n_out) tr_update(&ts, tab);
            if (threadIdx.x == 0) dst[i] = ts.out[ts.out_pos++];
            __syncthreads();
        }
        tr_store(t, &ts);
        return;
    }
    u64 field = 0; u32 cur_bit = 63, i = 0, j = 0; u64 a = 0;  // cur_bit = 63 forces a fetch on first use
    while (i < n) {                                            // every lane runs the same bookkeeping
        if (cur_bit == 63) {
            if (ts.out_pos >= ts.n_out) tr_update(&ts, tab);
            field = ts.out[ts.out_pos];
            __syncthreads();
            if (threadIdx.x == 0) ts.out_pos++;
            __syncthreads();
            cur_bit = 0;
        }
        if ((field >> cur_bit) & 1) a += 1ull << j;
        ++cur_bit; ++j;
        if (j == bits) { if (threadIdx.x == 0) dst[i] = a; ++i; a = 0; j = 0; }
    }
    tr_store(t, &ts);
}

// put(src[0..n_put)) then squeeze n_get words (or n_get indices of `bits` bits) in ONE launch: a proof's Fiat-Shamir steps are "absorb a root,
// draw one or two challenges" -- three launches of ~5 us floor each around a single permutation
__global__ __launch_bounds__(64) void tr_put_get_kernel(TranscriptState* t, const u64* __restrict__ src, u64 n_put, u64* __restrict__ dst, u32 n, u32 bits) {
    ZK_POSEIDON_LDS;
    __shared__ TranscriptState ts;
    load_tables(tab);
    tr_load(&ts, t);
    for (u64 i = 0; i < n_put; ++i) {                                       // tr_put_kernel
        if (threadIdx.x == 0) { ts.n_out = 0; ts.out_pos = 0; ts.pending[ts.n_pending++] = src[i]; }
        __syncthreads();
        if (ts.n_pending == 8) tr_update(&ts, tab);
    }
    if (bits == 0) {                                                        // tr_get_kernel
        for (u32 i = 0; i < n; ++i) {
            if (ts.out_pos >= ts.n_out) tr_update(&ts, tab);
            if (threadIdx.x == 0) dst[i] = ts.out[ts.out_pos++];
            __syncthreads();
        }
        tr_store(t, &ts);
        return;
    }
    u64 field = 0; u32 cur_bit = 63, i = 0, j = 0; u64 a = 0;
    while (i < n) {
        if (cur_bit == 63) {
            if (ts.out_pos >= ts.n_out) tr_update(&ts, tab);
            field = ts.out[ts.out_pos];
            __syncthreads();
            if (threadIdx.x == 0) ts.out_pos++;
            __syncthreads();
            cur_bit = 0;
        }
        if ((field >> cur_bit) & 1) a += 1ull << j;
        ++cur_bit; ++j;
        if (j == bits) { if (threadIdx.x == 0) dst[i] = a; ++i; a = 0; j = 0; }
    }
    tr_store(t, &ts);
}

// the coefficient matrix of one matrix-pipe product of the permutation (row-major [n_out][n_in]); which: MT_P, MT_BS + b, MT_BE + b
void mfma_coefficients(int which, u64 (&coef)[144], int& n_out, int& n_in) {
    if (which == MT_P) {
        n_out = n_in = 12;
        for (int o = 0; o < 12; ++o)
            for (int j = 0; j < 12; ++j) coef[o * 12 + j] = ZK_POSEIDON_P[12 * j + o];                   // out[o] = sum_j P[j][o] st[j]
    } else if (which < MT_BE) {
        const int b = which - MT_BS; n_out = PR_B; n_in = 11;
        for (int m = 0; m < PR_B; ++m)                                                                    // G[m] = sum_(k >= 1) S_(r0 + m)[k] st[k]
            for (int j = 0; j < 11; ++j) coef[m * 11 + j] = ZK_POSEIDON_S[23 * (PR_B * b + m) + 1 + j];
    } else {
        const int b = which - MT_BE; n_out = 11; n_in = PR_B;
        for (int o = 0; o < 11; ++o)                                                                      // st[o + 1] += sum_m SC_(r0 + m)[o + 1] u_m
            for (int m = 0; m < PR_B; ++m) coef[o * PR_B + m] = ZK_POSEIDON_S[23 * (PR_B * b + m) + 12 + o];
    }
}
// g_tab's image -> the one-lane kernels' image; "" or what went wrong.  Host only (also behind zk_poseidon_tables_selfcheck, no GPU needed).
std::string build_one_lane_image(const u64* tab, u64* mt) {
    for (int i = 0; i < TM_WORDS; ++i) mt[i] = 0;
    for (int i = 0; i < TM_S0; ++i) mt[i] = tab[i];                                                       // C0, FC, PC
    for (int r = 0; r < 22; ++r) { mt[TM_S0 + 2 * r] = tab[T_SR + 24 * r]; mt[TM_S0 + 2 * r + 1] = tab[T_SR + 24 * r + 1]; }
    for (int i = 0; i < 220; ++i) mt[TM_DD + i] = tab[T_DD + i];
    for (int i = 0; i < 12; ++i) {                                                                        // a zero word after round 0's S-box: C[i]^7 + C[12 + i]
        const u64 c = ZK_POSEIDON_C[i], c2 = gl::hmul(c, c), c3 = gl::hmul(c2, c), c7 = gl::hmul(gl::hmul(c3, c3), c), k = ZK_POSEIDON_C[12 + i];
        mt[TM_K0 + i] = c7 + k >= GL_P || c7 + k < c7 ? c7 + k - GL_P : c7 + k;
    }
    for (int which = 0; which < MT_N; ++which) {
        u64 coef[144]; int n_out, n_in;
        mfma_coefficients(which, coef, n_out, n_in);
        if (!pmfma::build_tables(coef, n_out, n_in, nullptr, mt + tm_table(which))) return "a digit column of product " + std::to_string(which) + " exceeds its bound";
        const std::string why = pmfma::check_tables(coef, n_out, n_in, nullptr, mt + tm_table(which));
        if (!why.empty()) return "product " + std::to_string(which) + ": " + why;
    }
    return "";
}

bool g_consts_loaded[64] = {};

std::mutex g_consts_mu;
// the cooperative kernels' LDS image (the T_ layout) from the constants of poseidon_gl_constants.h; host only
void build_coop_image(u64* tab) {
    // regroup C[118] by use (poseidon_opt.rs:98-199): initial add, post-S-box constants of the 8 full
    // rounds (C[12(R+1)+i] for R < 4, C[82+12(R-4)+i] for R = 4..6, none for the last), partial rounds
    for (int i = 0; i < T_WORDS; ++i) tab[i] = 0;
    for (int i = 0; i < 12; ++i) tab[T_C0 + i] = ZK_POSEIDON_C[i];
    for (int R = 0; R < 7; ++R)
        for (int i = 0; i < 12; ++i) tab[T_FC + 12 * R + i] = ZK_POSEIDON_C[(R < 4 ? 12 * (R + 1) : 82 + 12 * (R - 4)) + i];
    for (int r = 0; r < 22; ++r) tab[T_PC + r] = ZK_POSEIDON_C[60 + r];
    for (int i = 8; i < 12; ++i) {                        // a zero capacity word after round 0's S-box: C[i]^7 + C[12 + i]
        const u64 c = ZK_POSEIDON_C[i], c2 = gl::hmul(c, c), c3 = gl::hmul(c2, c), c7 = gl::hmul(gl::hmul(c3, c3), c), k = ZK_POSEIDON_C[12 + i];
        tab[T_K0 + i - 8] = c7 + k >= GL_P || c7 + k < c7 ? c7 + k - GL_P : c7 + k;
    }
    auto split = [&](int at, u64 c) { tab[at] = (c & 0x3FFFFF) | (((c >> 22) & 0x3FFFFF) << 32); tab[at + 1] = c >> 44; };
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) split(T_PT + 2 * (12 * i + j), ZK_POSEIDON_P[12 * j + i]);
    for (int r = 0; r < 22; ++r) {
        for (int j = 0; j < 12; ++j) split(T_SR + 2 * (12 * r + j), ZK_POSEIDON_S[23 * r + j]);
    }
    for (int b = 0; b < 22 / PR_B; ++b)                  // tables of partial_rounds()
        for (int m = 0; m < PR_B; ++m) {
            const int r = PR_B * b + m;
            for (int i = 0; i < m; ++i) {                // D[r][r0 + i] = sum_k S_r[k] SC_{r0 + i}[k]
                u64 d = 0;
                for (int k = 1; k < 12; ++k) {
                    const u64 t = gl::hmul(ZK_POSEIDON_S[23 * r + k], ZK_POSEIDON_S[23 * (PR_B * b + i) + 11 + k]);
                    d = d + t >= GL_P || d + t < d ? d + t - GL_P : d + t;
                }
                split(T_DD + 2 * ((PR_B * (PR_B - 1) / 2) * b + m * (m - 1) / 2 + i), d);
                split(T_CD + 2 * (16 * (PR_B * b + i) + m), d);      // round r0 + i's S-box output, in lane m > i
            }
            split(T_CD + 2 * (16 * r + m), ZK_POSEIDON_S[23 * r]);   // ... and in lane m of its own round: S_r[0]
            for (int k = 1; k < 12; ++k) split(T_SCS + 2 * (11 * PR_B * b + PR_B * (k - 1) + m), ZK_POSEIDON_S[23 * r + 11 + k]);
        }
}
void ensure_constants() {
    int dev; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    std::lock_guard<std::mutex> lk(g_consts_mu);          // provers on several host threads
    if (g_consts_loaded[dev]) return;
    static u64 tab[T_WORDS];
    build_coop_image(tab);
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), tab, sizeof(tab)));
    {   // LDS image of the one-lane kernels: their dense products run on the matrix pipe (gl_mfma.hip.h)
        static u64 mt[TM_WORDS];
        const std::string why = build_one_lane_image(tab, mt);
        ZK_REQUIRE(why.empty(), "Poseidon matrix-pipe tables: " + why);
        ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mtab), mt, sizeof(mt)));
    }
    g_consts_loaded[dev] = true;
}

}  // namespace

// Host-side self check of the one-lane kernels' tables (no GPU): the digit tables against their coefficients, and the matrix-pipe
// arithmetic step by step (gl_mfma.hip.h emulate_product) against 128-bit arithmetic on random and extreme vectors.  "" or what is wrong.
std::string poseidon_tables_selfcheck() {
    static u64 tab[T_WORDS], mt[TM_WORDS];
    build_coop_image(tab);
    const std::string why = build_one_lane_image(tab, mt);
    if (!why.empty()) return why;
    u64 seed = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { seed += 0x9E3779B97F4A7C15ull; u64 z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (int which = 0; which < MT_N; ++which) {
        u64 coef[144]; int n_out, n_in;
        mfma_coefficients(which, coef, n_out, n_in);
        for (int trial = 0; trial < 64; ++trial) {
            u64 x[12], got[12];
            for (int j = 0; j < 12; ++j)
                x[j] = trial == 0 ? 0 : trial == 1 ? ~0ull : trial == 2 ? GL_P - 1 : trial == 3 ? 0x8080808080808080ull : trial == 4 ? 0x7F7F7F7F7F7F7F7Full : rnd();
            pmfma::emulate_product(mt + tm_table(which), x, n_in, got);
            for (int o = 0; o < n_out; ++o) {
                unsigned __int128 want = 0;
                for (int j = 0; j < n_in; ++j) want = (want + (unsigned __int128)(coef[o * n_in + j] % GL_P) * (x[j] % GL_P)) % GL_P;
                if (got[o] != (u64)want) return "product " + std::to_string(which) + ", output " + std::to_string(o) + ", trial " + std::to_string(trial) + ": the matrix-pipe arithmetic disagrees with 128-bit arithmetic";
            }
        }
    }
    return "";
}

size_t transcript_state_bytes() { return sizeof(TranscriptState); }
void transcript_init_dev(void* d_t, hipStream_t st) {
    ensure_constants();
    hipLaunchKernelGGL(tr_init_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t);
    ZK_HIP(hipGetLastError());
}
void transcript_put_dev(void* d_t, const u64* d_src, uint64_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(tr_put_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, d_src, n);
    ZK_HIP(hipGetLastError());
}
void transcript_get_dev(void* d_t, u64* d_dst, uint32_t n_words, hipStream_t st) {
    hipLaunchKernelGGL(tr_get_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, d_dst, n_words, 0u);
    ZK_HIP(hipGetLastError());
}
void transcript_put_get_dev(void* d_t, const u64* d_src, uint64_t n_put, u64* d_dst, uint32_t n_get, uint32_t bits, hipStream_t st) {
    ZK_REQUIRE(bits <= 63, "get_permutations: nbits out of range");
    hipLaunchKernelGGL(tr_put_get_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, d_src, n_put, d_dst, n_get, bits);
    ZK_HIP(hipGetLastError());
}
void transcript_permutations_dev(void* d_t, uint32_t n, uint32_t nbits, u64* d_dst, hipStream_t st) {
    ZK_REQUIRE(nbits >= 1 && nbits <= 63, "get_permutations: nbits out of range");
    hipLaunchKernelGGL(tr_get_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, d_dst, n, nbits);
    ZK_HIP(hipGetLastError());
}

void poseidon_dev(const u64* d_in8, const u64* d_cap4, u64* d_out, int n_out, hipStream_t st) {
    ensure_constants();
    hipLaunchKernelGGL(poseidon_one_kernel, dim3(1), dim3(64), 0, st, d_in8, d_cap4, d_out, n_out);
    ZK_HIP(hipGetLastError());
}

void linearhash_rows_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st) {
    ensure_constants();
    if (height == 0) return;
    static const u64 coop_below = getenv("ZK_LH_COOP_BELOW") ? strtoull(getenv("ZK_LH_COOP_BELOW"), nullptr, 10) : 16384;
    static const u64 batch_upto = getenv("ZK_LH_BATCH_UPTO") ? strtoull(getenv("ZK_LH_BATCH_UPTO"), nullptr, 10) : 262144;
    const u32 bs = std::max<u32>(8, (width + 3) / 4), hsz = width > 4 ? (width + bs - 1) / bs : 1;
    static const u64 wave_below = getenv("ZK_LH_WAVE_BELOW") ? strtoull(getenv("ZK_LH_WAVE_BELOW"), nullptr, 10) : 4096;
    if (width <= 4) {
        hipLaunchKernelGGL(linearhash_pad_kernel, dim3((u32)((4 * height + 255) / 256)), dim3(256), 0, st, d_rows, width, height, d_digests);
    } else if (height < wave_below && hsz > 1) {      // fewest rows: a wave per row, its batches side by side
        hipLaunchKernelGGL(linearhash_rows_wave_kernel, dim3((u32)((height + 3) / 4)), dim3(256), 0, st, d_rows, width, height, d_digests);
    } else if (height < coop_below && width > 4) {    // few rows: latency-bound, 16 lanes per row
        hipLaunchKernelGGL(linearhash_rows_coop_kernel, dim3((u32)((height + 15) / 16)), dim3(256), 0, st, d_rows, width, height, d_digests);
    } else if (hsz > 1 && height <= batch_upto) {   // too few rows to fill the chip: the batches of a row side by side
        DevBuf h; h.reserve(height * hsz * 32);
        const u32 n_hashed = width - (hsz - 1) * bs <= 4 ? hsz - 1 : hsz;
        hipLaunchKernelGGL(linearhash_batch_kernel, dim3((u32)((height * n_hashed + ONE_THREADS - 1) / ONE_THREADS)), dim3(ONE_THREADS), 0, st, d_rows, width, height, bs, hsz, n_hashed, h.u());
        ZK_HIP(hipGetLastError());
        hipLaunchKernelGGL(linearhash_final_kernel, dim3((u32)((height + ONE_THREADS - 1) / ONE_THREADS)), dim3(ONE_THREADS), 0, st, (const u64*)h.u(), hsz, height, d_digests);
    } else {
        hipLaunchKernelGGL(linearhash_rows_kernel, dim3((u32)((height + ONE_THREADS - 1) / ONE_THREADS)), dim3(ONE_THREADS), 0, st, d_rows, width, height, d_digests);
    }
    ZK_HIP(hipGetLastError());
}

void merkle_roots_from_paths_dev(const u64* d_leaves, const u64* d_paths, const u32* d_depth, const u64* d_idx, uint32_t n, uint32_t max_depth,
                                 u64* d_roots, hipStream_t st) {
    ensure_constants();
    if (n == 0) return;
    hipLaunchKernelGGL(merkle_root_from_path_kernel, dim3((n + 3) / 4), dim3(64), 0, st, d_leaves, d_paths, d_depth, d_idx, n, max_depth, d_roots);
    ZK_HIP(hipGetLastError());
}

uint64_t merkle_n_nodes(uint64_t n_) {  // merklehash.rs:47-61
    uint64_t n = n_, next_n = (n - 1) / 2 + 1, acc = next_n * 2;
    while (n > 1) {
        n = next_n; next_n = (n - 1) / 2 + 1;
        if (n > 1) acc += next_n * 2; else acc += 1;
    }
    return acc;
}

void merkelize_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st) {
    ensure_constants();
    ZK_REQUIRE(height >= 1, "merkelize: height must be >= 1");
    const uint64_t nn = merkle_n_nodes(height);
    // absent right siblings on odd levels are the all-zero digest (merklehash.rs:307); a power-of-two height has no odd level and
    // every node is written below (a zero-width tree keeps its all-zero leaves from the clearing)
    if (width == 0 || height < 2 || (height & (height - 1))) ZK_HIP(hipMemsetAsync(d_nodes, 0, nn * 32, st));
    if (width == 0 && (height & (height - 1)) == 0 && height > 1) {  // all-zero leaves, full binary tree
        uint32_t levels = 0;
        while ((1ull << levels) < height) ++levels;
        // digest of an all-zero subtree of every height up to 2^32: a property of the hash, not of the proof --
        // computed once per device (33 serial permutations, 3 ms) and kept
        static u64* g_zero_chain[64] = {};
        static std::mutex g_zero_chain_mu;
        int dev; ZK_HIP(hipGetDevice(&dev));
        ZK_REQUIRE(levels <= 32, "merkelize: tree too tall");
        {   // provers on several host threads, each on its own stream: the chain is built once, under the lock, and is in memory
            // before its address is published (a second prover's fill kernels run on another stream and would not wait for this one)
            std::lock_guard<std::mutex> lk(g_zero_chain_mu);
            if (!g_zero_chain[dev]) {
                u64* d = nullptr;
                ZK_HIP(hipMalloc((void**)&d, 33 * 32));
                hipLaunchKernelGGL(zero_tree_chain_kernel, dim3(1), dim3(64), 0, st, 32u, d);
                ZK_HIP(hipGetLastError());
                ZK_HIP(hipStreamSynchronize(st));
                g_zero_chain[dev] = d;
            }
        }
        const u64* d_h = g_zero_chain[dev];
        uint64_t n = height, off = 0;
        for (uint32_t l = 1; l <= levels; ++l) {  // level l has height >> l nodes, starting after level l-1
            off += n; n >>= 1;
            hipLaunchKernelGGL(fill_digest_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, st, d_nodes + 4 * off, n, d_h + 4 * l);
            ZK_HIP(hipGetLastError());
        }
        return;
    }
    linearhash_rows_dev(d_rows, width, height, d_nodes, st);
    uint64_t n64 = height, next = (n64 - 1) / 2 + 1, p_in = 0, p_out = next * 2;
    while (n64 > 1) {  // merklehash.rs:331-343
        if (n64 <= 64) {      // the last six levels: one block, one launch (from 128 children the block's 16 waves crowd one CU: slower)
            const unsigned threads = (unsigned)std::min<uint64_t>(1024, std::max<uint64_t>(64, ((n64 + 1) / 2) * 16));
            hipLaunchKernelGGL(merkle_top_coop_kernel, dim3(1), dim3(threads), 0, st, d_nodes, n64, p_in);
            ZK_HIP(hipGetLastError());
            break;
        }
        static const u64 coop_upto = getenv("ZK_MERKLE_COOP_UPTO") ? strtoull(getenv("ZK_MERKLE_COOP_UPTO"), nullptr, 10) : 32768;
        if (next <= coop_upto) {  // few parents: latency-bound, 16 lanes per permutation
            hipLaunchKernelGGL(merkle_level_coop_kernel, dim3((u32)((next + 15) / 16)), dim3(256), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        } else {
            hipLaunchKernelGGL(merkle_level_kernel, dim3((u32)((next + ONE_THREADS - 1) / ONE_THREADS)), dim3(ONE_THREADS), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        }
        ZK_HIP(hipGetLastError());
        n64 = next; next = (n64 - 1) / 2 + 1; p_in = p_out; p_out = p_in + next * 2;
    }
}

}  // namespace zk
