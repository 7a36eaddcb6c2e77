// Per-stage glue kernels of the starky prover for gfx950: FRI folding and transposition, the
// x/(x-xi) tables, evaluation (opening) dot products, quotient split, vanishing-inverse and
// domain tables.  All operands are device resident; F3G = 3 consecutive canonical u64.
//
// Replaces (file:line under /root/reference/starky/src):
//   fri.rs:101-126 (fold, single-threaded `for g` there), fri.rs:299-317 (get_transposed_buffer),
//   stark_gen.rs:231-249 (x_n, x_2ns, build_Zh_Inv :575-592), :375-396 (Q split),
//   :416-430 (LEv/LpEv), :432-466 (evals), :481-522 (xDivXSubXi via sequential batch_inverse,
//   polutils.rs:35-53 -- here every point is inverted independently: same field elements).
#include "zk_internal.h"
#include "acc6.hip.h"
#include <vector>
#include <cstring>
#include <algorithm>
#include "ntt_reg.hip.h"

namespace zk {

namespace {

using gl::f3;

__device__ __forceinline__ f3 ld3(const u64* __restrict__ p) { return f3{{p[0], p[1], p[2]}}; }
__device__ __forceinline__ void st3(u64* __restrict__ p, f3 v) { p[0] = v.v[0]; p[1] = v.v[1]; p[2] = v.v[2]; }
__device__ __forceinline__ f3 f3_pow(f3 a, u64 e) {
    f3 r{{1, 0, 0}};
    while (e) { if (e & 1) r = gl::f3_mul(r, a); a = gl::f3_mul(a, a); e >>= 1; }
    return r;
}
// multiplication by the generator x of GF(p^3) = GF(p)[x]/(x^3 - x - 1)
__device__ __forceinline__ f3 mul_x(f3 a) { return f3{{a.v[2], gl::add(a.v[0], a.v[2]), a.v[1]}}; }

// ---- FRI fold (fri.rs:112-126) ------------------------------------------------------------
// pol2[g] = P_g(special_x * sinv_g), P_g = interpolant of {pol[i*n2 + g]}_i on <w_NX>,
// sinv_g = shift_inv * w_polbits^-g.  One lane per g; limb by limb: a base-field size-NX inverse
// DFT in registers, Horner at the extension point, recombination with powers of x.
template <int LOGNX>
__global__ __launch_bounds__(256) void fri_fold_kernel(const u64* __restrict__ pol, u64 pol2_n, const u64* __restrict__ w256inv,
                                                       u64 shift_inv, u64 wi, u64 nx_inv, const u64* __restrict__ special_x,
                                                       u64* __restrict__ out) {
    // Four lanes per g, one per limb (the fourth idles): the three limb polynomials are independent until the last line, and a
    // fold has few outputs (2^4 ... 2^19) with a long serial chain each (NX - 1 extension products per limb).
    constexpr int NX = 1 << LOGNX;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 g0 = t >> 2, g = g0 < pol2_n ? g0 : pol2_n - 1;          // idle quads shadow the last one (the shuffles below need every lane)
    const int l = (int)(t & 3) < 3 ? (int)(t & 3) : 2;
    const u64 sinv = gl::mul(shift_inv, gl::pow(wi, g));
    const f3 y = gl::f3_muls(ld3(special_x), sinv);
    u64 x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = pol[((u64)i * pol2_n + g) * 3 + l];
    ntt_reg<LOGNX, true>(x);  // coefficient k (times NX) sits in x[bitrev(k)]
    f3 acc{{x[bitrev_c(NX - 1, LOGNX)], 0, 0}};
#pragma unroll
    for (int k = NX - 2; k >= 0; --k) {
        acc = gl::f3_mul(acc, y);
        acc.v[0] = gl::add(acc.v[0], x[bitrev_c(k, LOGNX)]);
    }
    f3 S1, S2;                                                          // lane 0 of the quad collects the other two limbs' sums
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        S1.v[j] = ((u64)(u32)__shfl((int)(u32)(acc.v[j] >> 32), 1, 4) << 32) | (u32)__shfl((int)(u32)acc.v[j], 1, 4);
        S2.v[j] = ((u64)(u32)__shfl((int)(u32)(acc.v[j] >> 32), 2, 4) << 32) | (u32)__shfl((int)(u32)acc.v[j], 2, 4);
    }
    if ((t & 3) || g0 >= pol2_n) return;
    const f3 r = gl::f3_add(acc, gl::f3_add(mul_x(S1), mul_x(mul_x(S2))));
    st3(out + 3 * g, gl::f3_muls(r, nx_inv));
}

// The same fold for reductions of more than 6 bits per step (final.starkStruct.*.json folds 2^17 -> 2^7: 1024 points per
// group): one workgroup per g, the size-NX inverse transform of all three limbs in LDS (radix-2 decimation in time over
// bit-reversed loads), then the evaluation at y as 256 per-thread Horner chunks folded pairwise with y^CH, y^2CH, ...
// Only the last, small FRI steps come here, so clarity wins over speed.
constexpr int FOLD_BIG_MAX_LOG = 11;
__global__ __launch_bounds__(256) void fri_fold_big_kernel(const u64* __restrict__ pol, u64 pol2_n, u32 lognx, u64 shift_inv, u64 wi, u64 w_nx_inv,
                                                           u64 nx_inv, const u64* __restrict__ special_x, u64* __restrict__ out) {
    extern __shared__ u64 fold_lds[];
    const u32 NX = 1u << lognx, t = threadIdx.x;
    u64* buf = fold_lds;                 // [NX][3]
    u64* tw = fold_lds + 3 * (u64)NX;    // [NX/2]: w_nx^-j
    const u64 g = blockIdx.x;
    for (u32 i = t; i < NX; i += 256) {
        const u32 rv = __brev(i) >> (32 - lognx);
        const f3 v = ld3(pol + ((u64)i * pol2_n + g) * 3);
        st3(buf + 3 * (u64)rv, v);
    }
    for (u32 j = t; j < NX / 2; j += 256) tw[j] = gl::pow(w_nx_inv, j);
    __syncthreads();
    for (u32 s = 1; s <= lognx; ++s) {
        const u32 half = 1u << (s - 1), stride = NX >> s;
        for (u32 b = t; b < NX / 2; b += 256) {
            const u32 j = b & (half - 1), k = ((b >> (s - 1)) << s) + j;
            const f3 u = ld3(buf + 3 * (u64)k), v = gl::f3_muls(ld3(buf + 3 * (u64)(k + half)), tw[j * stride]);
            st3(buf + 3 * (u64)k, gl::f3_add(u, v));
            st3(buf + 3 * (u64)(k + half), gl::f3_sub(u, v));
        }
        __syncthreads();
    }
    // buf[k] = NX * c_k.  P(y) = sum_k c_k y^k with y = special_x * shift_inv * wi^g
    const f3 y = gl::f3_muls(ld3(special_x), gl::mul(shift_inv, gl::pow(wi, g)));
    const u32 CH = NX / 256 ? NX / 256 : 1;                       // coefficients per thread (NX >= 128 here; fewer threads work when NX < 256)
    const u32 n_part = NX / CH;
    f3 acc{{0, 0, 0}};
    if (t < n_part) {
        acc = ld3(buf + 3 * (u64)(t * CH + CH - 1));
        for (int j = (int)CH - 2; j >= 0; --j) acc = gl::f3_add(gl::f3_mul(acc, y), ld3(buf + 3 * (u64)(t * CH + j)));
    }
    __syncthreads();
    if (t < n_part) st3(buf + 3 * (u64)t, acc);
    f3 Y = f3_pow(y, CH);
    __syncthreads();
    for (u32 n = n_part; n > 1; n >>= 1) {                        // p[i] = p[2i] + p[2i+1] * Y, Y <- Y^2
        f3 r{{0, 0, 0}};
        if (t < n / 2) r = gl::f3_add(ld3(buf + 3 * (u64)(2 * t)), gl::f3_mul(ld3(buf + 3 * (u64)(2 * t + 1)), Y));
        __syncthreads();
        if (t < n / 2) st3(buf + 3 * (u64)t, r);
        Y = gl::f3_mul(Y, Y);
        __syncthreads();
    }
    if (t == 0) st3(out + 3 * g, gl::f3_muls(ld3(buf), nx_inv));
}

__global__ void copy3_kernel(const u64* __restrict__ in, u64 n, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// fri.rs:299-317: out[(i*h + j)*3 ..] = pol[(j*w + i)*3 ..]
__global__ void fri_transpose_kernel(const u64* __restrict__ pol, u64 n, u32 tbits, u64* __restrict__ out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;  // destination element index
    if (t >= n) return;
    const u64 w = 1ull << tbits, h = n >> tbits;
    const u64 i = t / h, j = t % h;
    st3(out + 3 * t, ld3(pol + 3 * (j * w + i)));
}

// ---- domain tables -------------------------------------------------------------------------
__global__ void x_table_kernel(u64 shift, u64 w, u64 n, u64* __restrict__ out) {   // stark_gen.rs:231-247
    const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = gl::mul(shift, gl::pow(w, k));
}
__global__ void zh_inv_kernel(u64 sn, u64 we, u32 n, u64* __restrict__ out) {      // stark_gen.rs:575-592
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = gl::inv(gl::sub(gl::mul(sn, gl::pow(we, j)), 1));
}
// x/(x - xi*mulw) for x = 49*w_ext^k                                            stark_gen.rs:481-522
// Eight consecutive points per lane: one exponentiation for the first x, one extension-field inversion for all
// eight denominators (Montgomery's trick: prefix products, invert the last, walk back).
constexpr int XD_PER = 8;
// (blockIdx.y picks one of two multipliers / outputs: x/(x - xi) and x/(x - xi w) of a proof are one launch -- each is a latency-bound
// chain of an exponentiation and an inversion per lane, and two launches stood one behind the other)
__global__ __launch_bounds__(256) void xdivxsub_kernel(const u64* __restrict__ xi, u64 mulw0, u64 mulw1, u64 w_ext, u64 n, u64* __restrict__ out0, u64* __restrict__ out1) {
    const u64 k0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * XD_PER;
    if (k0 >= n) return;
    const u64 mulw = blockIdx.y ? mulw1 : mulw0;
    u64* __restrict__ out = blockIdx.y ? out1 : out0;
    const f3 z = gl::f3_muls(ld3(xi), mulw);
    u64 x[XD_PER]; f3 den[XD_PER], pre[XD_PER];
    x[0] = gl::mul(49, gl::pow(w_ext, k0));
#pragma unroll
    for (int j = 1; j < XD_PER; ++j) x[j] = gl::mul(x[j - 1], w_ext);
#pragma unroll
    for (int j = 0; j < XD_PER; ++j) {
        den[j] = f3{{gl::sub(x[j], z.v[0]), gl::neg(z.v[1]), gl::neg(z.v[2])}};
        pre[j] = j ? gl::f3_mul(pre[j - 1], den[j]) : den[0];
    }
    f3 inv = gl::f3_inv(pre[XD_PER - 1]);
#pragma unroll
    for (int j = XD_PER - 1; j >= 0; --j) {
        const f3 dj = j ? gl::f3_mul(inv, pre[j - 1]) : inv;      // 1/den[j]
        if (j) inv = gl::f3_mul(inv, den[j]);
        if (k0 + j < n) st3(out + 3 * (k0 + j), gl::f3_muls(dj, x[j]));
    }
}
// LEv[i] = (xi * mulw / 49)^i                                                   stark_gen.rs:416-427
struct LevPowArgs { u64 c[4]; u64* out[4]; };           // up to four tables of powers in one launch (blockIdx.y): LEv, LpEv, the quotient's weights
__global__ __launch_bounds__(256) void lev_pow_kernel(const u64* __restrict__ xi, const LevPowArgs A, u64 n) {
    const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * XD_PER;   // eight consecutive powers per lane
    if (i0 >= n) return;
    const u64 mul_c = A.c[blockIdx.y];
    u64* __restrict__ out = A.out[blockIdx.y];
    const f3 b = gl::f3_muls(ld3(xi), mul_c);
    f3 p = f3_pow(b, i0);
#pragma unroll
    for (int j = 0; j < XD_PER; ++j) {
        if (i0 + j < n) st3(out + 3 * (i0 + j), p);
        p = gl::f3_mul(p, b);
    }
}

// ---- evals (stark_gen.rs:432-466) ----------------------------------------------------------
struct EvalDesc { const u64* buf; u64 width; u64 offset; u32 dim; u32 prime; };
// the descriptor the kernel sees: which weight table (LEv / LpEv of the caller), and the row stride of the section as a shift
struct EvalDescK { const u64* buf; const u64* L; u64 width; u64 offset; u32 dim; u32 rshift; };
constexpr int EV_BLOCKS = 4096;  // blocks along the row axis: the row loop is latency-bound, more waves in flight = more loads in flight
                                 // (512 blocks: 7.1 ms per 2^24-row proof, 2048: 5.5 ms)
constexpr int EV_LANES = 32;     // evaluations per block: consecutive descriptors sit in consecutive lanes
struct EvalBatch { EvalDescK d[EV_LANES]; u32 out[EV_LANES]; };   // 32 base-field columns (+ their partial-sum slots) travel as a kernel argument: no upload, no wait for one
constexpr int EV_UNROLL = 4;     // rows per lane and trip, all loads issued before the first product
constexpr int EV_FLUSH = 256;    // terms an Acc6 may take before it is folded (acc6.hip.h: n * 2^54 < 2^64)

// One block = 8 x EV_UNROLL rows x 32 columns per trip: the 32 lanes of a row read neighbouring columns of the same section row
// (ev_map lists a section's columns one after the other), so a trip reads whole row segments; L[k] is shared by the lanes of a row.
// slot = sum_k section[(k << rshift)][offset] * L[k] over k < N, a base-field column against extension-field weights.
// Measured (rocprofv3, 2^24 rows): the first version -- a modular product and a modular addition per limb of L[k], ~100
// instructions per (column, row) -- took 1.7 ms per 32 columns whatever the loads did, and an extension column, alone in a launch
// of its own kind, 3.3 ms: the kernel is bound by wave-instructions.  Now a column word is split in 22/22/20-bit limbs once per row
// and the three limbs of L[k] accumulate carry-free (18 multiply-adds per row, one fold per EV_FLUSH rows), and an extension column
// is three base-field columns (the product by L[k] is linear in the column: c L = c0 L + x (c1 L) + x^2 (c2 L), combined at the end).
__global__ __launch_bounds__(256) void evals_partial_kernel(const EvalBatch B, u32 n_live, u32 nbits, u32 n_row_blocks,
                                                            u64* __restrict__ partial /* [slots][n_row_blocks][3] */) {
    const u32 lane = threadIdx.x % EV_LANES, rowl = threadIdx.x / EV_LANES;   // 8 rows per trip and unroll step
    const bool live = lane < n_live;
    const EvalDescK d = B.d[live ? lane : 0];
    const u64 N = 1ull << nbits;
    const u64* __restrict__ L = d.L;
    const u64 stride = (u64)gridDim.x * 8;
    f3 acc{{0, 0, 0}};
    Acc6 A[3];
    acc_zero(A[0]); acc_zero(A[1]); acc_zero(A[2]);
    u32 pending = 0;
    auto term = [&](u64 c, const f3& l) {
        const u32 c0 = (u32)c & 0x3FFFFFu, c1 = (u32)(c >> 22) & 0x3FFFFFu, c2 = (u32)(c >> 44);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const u32 x0 = (u32)l.v[i], x1 = (u32)(l.v[i] >> 32);
            A[i].a00 += (u64)c0 * x0; A[i].a10 += (u64)c1 * x0; A[i].a20 += (u64)c2 * x0;
            A[i].a01 += (u64)c0 * x1; A[i].a11 += (u64)c1 * x1; A[i].a21 += (u64)c2 * x1;
        }
    };
    auto fold = [&] {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            u64 r = acc_finish(A[i]);
            r = r >= GL_P ? r - GL_P : r;
            acc.v[i] = gl::add(acc.v[i], r);
            acc_zero(A[i]);
        }
        pending = 0;
    };
    if (live) {
        u64 k = (u64)blockIdx.x * 8 + rowl;
        for (; k + (EV_UNROLL - 1) * stride < N; k += EV_UNROLL * stride) {
            f3 l[EV_UNROLL]; u64 c[EV_UNROLL];
#pragma unroll
            for (int j = 0; j < EV_UNROLL; ++j) {
                const u64 kk = k + j * stride;
                l[j] = ld3(L + 3 * kk);
                c[j] = d.buf[(kk << d.rshift) * d.width + d.offset];
            }
#pragma unroll
            for (int j = 0; j < EV_UNROLL; ++j) term(c[j], l[j]);
            pending += EV_UNROLL;
            if (pending >= (u32)EV_FLUSH) fold();
        }
        for (; k < N; k += stride) { term(d.buf[(k << d.rshift) * d.width + d.offset], ld3(L + 3 * k)); if (++pending >= (u32)EV_FLUSH) fold(); }
        fold();
    }
    __shared__ u64 red[256 * 3];
    st3(red + threadIdx.x * 3, acc);
    __syncthreads();
    for (int s = 4; s > 0; s >>= 1) {   // over the 8 rows of the block
        if ((int)rowl < s) st3(red + threadIdx.x * 3, gl::f3_add(ld3(red + threadIdx.x * 3), ld3(red + (threadIdx.x + s * EV_LANES) * 3)));
        __syncthreads();
    }
    if (rowl == 0 && live) st3(partial + ((u64)B.out[lane] * n_row_blocks + blockIdx.x) * 3, ld3(red + lane * 3));
}
// out[e] = the sum of slot s over the row blocks for a base-field column, s + x (s + 1) + x^2 (s + 2) for an extension column
// (x (a0, a1, a2) = (a2, a0 + a2, a1) in GF(p)[x] / (x^3 - x - 1)).  One wave per evaluation; 64 evaluations per launch.
struct EvalFinalBatch { u32 slot[64]; u32 dim[64]; };
__global__ __launch_bounds__(64) void evals_final_kernel(const u64* __restrict__ partial, const EvalFinalBatch F, u32 e0, u32 nblk, u64* __restrict__ out) {
    const u32 e = blockIdx.x;
    __shared__ u64 red[64 * 3];
    f3 res{{0, 0, 0}};
    for (u32 j = 0; j < F.dim[e]; ++j) {
        f3 acc{{0, 0, 0}};
#pragma unroll 4
        for (u32 b = threadIdx.x; b < nblk; b += 64) acc = gl::f3_add(acc, ld3(partial + ((u64)(F.slot[e] + j) * nblk + b) * 3));
        st3(red + threadIdx.x * 3, acc);
        __syncthreads();
        for (int s = 32; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) st3(red + threadIdx.x * 3, gl::f3_add(ld3(red + threadIdx.x * 3), ld3(red + (threadIdx.x + s) * 3)));
            __syncthreads();
        }
        f3 t = ld3(red);
        __syncthreads();
        for (u32 m = 0; m < j; ++m) t = f3{{t.v[2], gl::add(t.v[0], t.v[2]), t.v[1]}};     // times x, j times
        res = gl::f3_add(res, t);
    }
    if (threadIdx.x == 0) st3(out + 3 * (e0 + e), res);
}

// ---- Q split (stark_gen.rs:375-391) ---------------------------------------------------------
__global__ void qsplit_kernel(const u64* __restrict__ qq1, u32 nbits, u32 q_dim, u32 q_deg, u64 shift_inv_n, u64* __restrict__ qq2) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;  // over N * q_deg * q_dim source cells
    const u64 N = 1ull << nbits;
    if (t >= N * q_deg * q_dim) return;
    const u32 k = t % q_dim;
    const u64 row = t / q_dim;          // p*N + i
    const u64 p = row >> nbits, i = row & (N - 1);
    qq2[i * q_dim * q_deg + q_dim * p + k] = gl::mul(qq1[t], gl::pow(shift_inv_n, p));
}

// ---- get_pol / set_pol (stark_gen.rs:594-622, :683-707): one column of a section <-> [n][3] ------
__global__ void pol_get_kernel(const u64* __restrict__ buf, u64 width, u64 offset, u32 dim, u64 n, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64* c = buf + i * width + offset;
    out[3 * i] = c[0]; out[3 * i + 1] = dim == 3 ? c[1] : 0; out[3 * i + 2] = dim == 3 ? c[2] : 0;
}
__global__ void pol_set_kernel(u64* __restrict__ buf, u64 width, u64 offset, u32 dim, u64 n, const u64* __restrict__ in) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64* c = buf + i * width + offset;
    c[0] = in[3 * i];
    if (dim == 3) { c[1] = in[3 * i + 1]; c[2] = in[3 * i + 2]; }
}

// ---- calculate_Z (stark_gen.rs:653-666): z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1] -----------
// The reference's sequential prefix product becomes a three-phase scan over chunks of 1024 rows;
// every den is inverted on its own (the reference's batch_inverse yields the same field elements).
constexpr int Z_ITEMS = 4, Z_CHUNK = 256 * Z_ITEMS;

__device__ __forceinline__ f3 block_scan_excl(f3 v, u64* lds /* 256*3 */, f3* total) {
    // Hillis-Steele inclusive scan of 256 products, returned as exclusive prefix
    const int t = threadIdx.x;
    st3(lds + 3 * t, v);
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        f3 cur = ld3(lds + 3 * t);
        f3 prev = t >= off ? ld3(lds + 3 * (t - off)) : f3{{1, 0, 0}};
        __syncthreads();
        st3(lds + 3 * t, gl::f3_mul(cur, prev));
        __syncthreads();
    }
    *total = ld3(lds + 3 * 255);
    f3 ex = t ? ld3(lds + 3 * (t - 1)) : f3{{1, 0, 0}};
    __syncthreads();
    return ex;
}

__global__ __launch_bounds__(256) void z_ratio_kernel(const u64* __restrict__ num, const u64* __restrict__ den, u64 n,
                                                      u64* __restrict__ ratio, u64* __restrict__ block_tot) {
    __shared__ u64 lds[256 * 3];
    const u64 base = (u64)blockIdx.x * Z_CHUNK + (u64)threadIdx.x * Z_ITEMS;
    f3 prod{{1, 0, 0}};
#pragma unroll
    for (int k = 0; k < Z_ITEMS; ++k) {
        const u64 i = base + k;
        if (i < n) {
            f3 r = gl::f3_mul(ld3(num + 3 * i), gl::f3_inv(ld3(den + 3 * i)));
            st3(ratio + 3 * i, r);
            prod = gl::f3_mul(prod, r);
        }
    }
    f3 total;
    block_scan_excl(prod, lds, &total);
    if (threadIdx.x == 0) st3(block_tot + 3 * blockIdx.x, total);
}
// exclusive scan of the chunk products, in place: lane t owns ceil(n_blocks / 256) consecutive chunks (one lane walking all of them
// was 0.8 us per chunk: 0.2 ms at 2^18 rows, 12 ms at 2^24)
__global__ __launch_bounds__(256) void z_block_offsets_kernel(u64* __restrict__ block_tot, u64 n_blocks) {
    __shared__ u64 lds[256 * 3];
    const u64 per = (n_blocks + 255) / 256;
    const u64 b0 = std::min<u64>((u64)threadIdx.x * per, n_blocks), b1 = std::min<u64>(b0 + per, n_blocks);
    f3 prod{{1, 0, 0}};
    for (u64 b = b0; b < b1; ++b) prod = gl::f3_mul(prod, ld3(block_tot + 3 * b));
    f3 total;
    f3 acc = block_scan_excl(prod, lds, &total);
    for (u64 b = b0; b < b1; ++b) { const f3 t = ld3(block_tot + 3 * b); st3(block_tot + 3 * b, acc); acc = gl::f3_mul(acc, t); }
    if (threadIdx.x == 0) st3(block_tot + 3 * n_blocks, total);  // grand product: must be 1 (stark_gen.rs:663-664)
}
__global__ __launch_bounds__(256) void z_apply_kernel(const u64* __restrict__ ratio, const u64* __restrict__ block_off, u64 n,
                                                      u64* __restrict__ z) {
    __shared__ u64 lds[256 * 3];
    const u64 base = (u64)blockIdx.x * Z_CHUNK + (u64)threadIdx.x * Z_ITEMS;
    f3 r[Z_ITEMS], prod{{1, 0, 0}};
#pragma unroll
    for (int k = 0; k < Z_ITEMS; ++k) {
        r[k] = base + k < n ? ld3(ratio + 3 * (base + k)) : f3{{1, 0, 0}};
        prod = gl::f3_mul(prod, r[k]);
    }
    f3 total;
    f3 acc = gl::f3_mul(block_scan_excl(prod, lds, &total), ld3(block_off + 3 * blockIdx.x));
#pragma unroll
    for (int k = 0; k < Z_ITEMS; ++k) {
        if (base + k < n) st3(z + 3 * (base + k), acc);
        acc = gl::f3_mul(acc, r[k]);
    }
}

inline dim3 grid1(u64 n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

void pol_get_dev(const u64* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n, u64* d_out, hipStream_t st) {
    ZK_REQUIRE(dim == 1 || dim == 3, "get_pol: Invalid dim");
    hipLaunchKernelGGL(pol_get_kernel, grid1(n), dim3(256), 0, st, d_buf, width, offset, dim, n, d_out);
    ZK_HIP(hipGetLastError());
}
void pol_set_dev(u64* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n, const u64* d_in, hipStream_t st) {
    ZK_REQUIRE(dim == 1 || dim == 3, "set_pol: Invalid dim");
    hipLaunchKernelGGL(pol_set_kernel, grid1(n), dim3(256), 0, st, d_buf, width, offset, dim, n, d_in);
    ZK_HIP(hipGetLastError());
}
// d_work: (n + n/1024 + 2) * 3 words of scratch; d_check receives the grand product (3 words)
void calculate_z_dev(const u64* d_num, const u64* d_den, uint64_t n, u64* d_z, u64* d_work, u64* d_check, hipStream_t st) {
    const u64 nb = (n + Z_CHUNK - 1) / Z_CHUNK;
    u64* ratio = d_work;
    u64* tot = d_work + 3 * n;
    hipLaunchKernelGGL(z_ratio_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_num, d_den, n, ratio, tot);
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(z_block_offsets_kernel, dim3(1), dim3(256), 0, st, tot, nb);
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(z_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, ratio, tot, n, d_z);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(d_check, tot + 3 * nb, 24, hipMemcpyDeviceToDevice, st));
}

void fri_fold_dev(const u64* d_pol, uint32_t pol_bits, uint32_t step_bits, const u64* d_special_x, u64 shift_inv, u64* d_out, hipStream_t st) {
    ZK_REQUIRE(step_bits <= pol_bits && pol_bits <= 32, "fri_fold: need step_bits <= pol_bits <= 32");
    const uint32_t r = pol_bits - step_bits;
    const u64 n2 = 1ull << step_bits;
    if (r == 0) {  // fri.rs:113-114: step 0 copies
        hipLaunchKernelGGL(copy3_kernel, grid1(n2 * 3), dim3(256), 0, st, d_pol, n2 * 3, d_out);
        ZK_HIP(hipGetLastError());
        return;
    }
    ZK_REQUIRE(r <= (uint32_t)FOLD_BIG_MAX_LOG, "fri_fold: reduction of more than 11 bits per step is not supported");
    const u64* w256inv = ntt_w256_table(true);
    const u64 wi = gl::hinv(gl::hroot(pol_bits)), nx_inv = gl::hinv(1ull << r);
    if (r > 6) {
        const size_t lds = ((size_t)3 << r) * 8 + ((size_t)1 << (r - 1)) * 8;
        hipLaunchKernelGGL(fri_fold_big_kernel, dim3((unsigned)n2), dim3(256), lds, st, d_pol, n2, r, shift_inv, wi, gl::hinv(gl::hroot(r)), nx_inv, d_special_x, d_out);
        ZK_HIP(hipGetLastError());
        return;
    }
#define ZK_FOLD(L) hipLaunchKernelGGL((fri_fold_kernel<L>), grid1(4 * n2), dim3(256), 0, st, d_pol, n2, w256inv, shift_inv, wi, nx_inv, d_special_x, d_out)
    switch (r) {
        case 1: ZK_FOLD(1); break; case 2: ZK_FOLD(2); break; case 3: ZK_FOLD(3); break;
        case 4: ZK_FOLD(4); break; case 5: ZK_FOLD(5); break; default: ZK_FOLD(6); break;
    }
#undef ZK_FOLD
    ZK_HIP(hipGetLastError());
}

void fri_transpose_dev(const u64* d_pol, uint64_t n, uint32_t tbits, u64* d_out, hipStream_t st) {
    ZK_REQUIRE((n >> tbits) >= 1, "fri_transpose: bad shape");
    hipLaunchKernelGGL(fri_transpose_kernel, grid1(n), dim3(256), 0, st, d_pol, n, tbits, d_out);
    ZK_HIP(hipGetLastError());
}

void x_table_dev(uint32_t nbits, u64 shift, u64* d_out, hipStream_t st) {
    const u64 n = 1ull << nbits;
    hipLaunchKernelGGL(x_table_kernel, grid1(n), dim3(256), 0, st, shift, gl::hroot(nbits), n, d_out);
    ZK_HIP(hipGetLastError());
}

void zh_inv_dev(uint32_t nbits, uint32_t extend_bits, u64* d_out, hipStream_t st) {
    u64 sn = 49;
    for (uint32_t i = 0; i < nbits; ++i) sn = gl::hmul(sn, sn);
    const uint32_t n = 1u << extend_bits;
    hipLaunchKernelGGL(zh_inv_kernel, grid1(n), dim3(256), 0, st, sn, gl::hroot(extend_bits), n, d_out);
    ZK_HIP(hipGetLastError());
}

void xdivxsub_dev(const u64* d_xi, u64 mulw, uint32_t nbits_ext, u64* d_out, hipStream_t st) {
    const u64 n = 1ull << nbits_ext;
    hipLaunchKernelGGL(xdivxsub_kernel, grid1((n + XD_PER - 1) / XD_PER), dim3(256), 0, st, d_xi, mulw, mulw, gl::hroot(nbits_ext), n, d_out, d_out);
    ZK_HIP(hipGetLastError());
}
void xdivxsub2_dev(const u64* d_xi, u64 mulw0, u64 mulw1, uint32_t nbits_ext, u64* d_out0, u64* d_out1, hipStream_t st) {   // both tables of a proof, one launch
    const u64 n = 1ull << nbits_ext;
    dim3 g = grid1((n + XD_PER - 1) / XD_PER); g.y = 2;
    hipLaunchKernelGGL(xdivxsub_kernel, g, dim3(256), 0, st, d_xi, mulw0, mulw1, gl::hroot(nbits_ext), n, d_out0, d_out1);
    ZK_HIP(hipGetLastError());
}

// LEv / LpEv (stark_gen.rs:416-430): d_pow <- powers of xi / shift (prime: xi w / shift), d_out <- their inverse transform.
// shift = 49 gives the reference's tables (weights of the rows k 2^ext of an extended section); shift = 1 the weights of the
// rows of the section itself, and d_pow then holds the powers of xi (the weights of a polynomial's coefficients).
void lev_pow_dev(const u64* d_xi, uint32_t nbits, bool prime, u64 shift, u64* d_pow, hipStream_t st) {   // (xi / shift)^k or (xi w / shift)^k, k < 2^nbits
    const u64 n = 1ull << nbits;
    u64 c = gl::hinv(shift);
    if (prime) c = gl::hmul(c, gl::hroot(nbits));
    LevPowArgs A; memset(&A, 0, sizeof A); A.c[0] = c; A.out[0] = d_pow;
    hipLaunchKernelGGL(lev_pow_kernel, grid1((n + XD_PER - 1) / XD_PER), dim3(256), 0, st, d_xi, A, n);
    ZK_HIP(hipGetLastError());
}
// several tables of powers in one launch: table v holds (xi w^prime[v] / shift[v])^k, k < 2^nbits
void lev_pow_multi_dev(const u64* d_xi, uint32_t nbits, uint32_t n_tables, const bool* prime, const u64* shift, u64* const* d_pow, hipStream_t st) {
    ZK_REQUIRE(n_tables >= 1 && n_tables <= 4, "lev_pow_multi: 1..4 tables");
    const u64 n = 1ull << nbits;
    LevPowArgs A; memset(&A, 0, sizeof A);
    for (u32 v = 0; v < n_tables; ++v) { u64 c = gl::hinv(shift[v]); if (prime[v]) c = gl::hmul(c, gl::hroot(nbits)); A.c[v] = c; A.out[v] = d_pow[v]; }
    dim3 g = grid1((n + XD_PER - 1) / XD_PER); g.y = n_tables;
    hipLaunchKernelGGL(lev_pow_kernel, g, dim3(256), 0, st, d_xi, A, n);
    ZK_HIP(hipGetLastError());
}
void lev_dev(const u64* d_xi, uint32_t nbits, bool prime, u64 shift, u64* d_out, u64* d_pow, u64* d_tmp2, hipStream_t st) {
    lev_pow_dev(d_xi, nbits, prime, shift, d_pow, st);
    ntt_dev(d_pow, d_out, d_tmp2, 3, nbits, true, st);  // FFT::ifft over F3G == per-limb iNTT (base-field roots)
}

// evaluations at xi (stark_gen.rs:432-466): out[e] = sum_k section_e[k << rshift_e] * L_e[k], k < 2^nbits
void evals_k_dev(const EvalDescKHost* descs, uint32_t n_ev, uint32_t nbits, u64* d_out, hipStream_t st) {
    if (n_ev == 0) return;
    static_assert(sizeof(EvalDescKHost) == sizeof(EvalDescK), "descriptor layout");
    u32 n_slots = 0;
    for (u32 e = 0; e < n_ev; ++e) { ZK_REQUIRE(descs[e].dim == 1 || descs[e].dim == 3, "evals: dim must be 1 or 3"); n_slots += descs[e].dim; }
    // row blocks: a block takes 8 x EV_UNROLL rows per trip; a small proof gets no more blocks than it has trips (2^10 rows:
    // 32 blocks instead of 4096 -- the final kernel then adds 32 partial sums per evaluation, not 4096)
    const u32 n_row_blocks = (u32)std::min<u64>(EV_BLOCKS, std::max<u64>(1, (1ull << nbits) / (8 * EV_UNROLL)));
    DevBuf partial;  // pooled block; returned to the pool at scope exit, reuse is stream ordered
    partial.reserve((size_t)n_slots * n_row_blocks * 24);
    EvalBatch B; memset(&B, 0, sizeof B);
    u32 n = 0, slot = 0;
    auto flush = [&] {
        if (!n) return;
        hipLaunchKernelGGL(evals_partial_kernel, dim3(n_row_blocks), dim3(256), 0, st, B, n, nbits, n_row_blocks, partial.u());
        n = 0;
    };
    std::vector<u32> slot0(n_ev);
    for (u32 e = 0; e < n_ev; ++e) {
        slot0[e] = slot;
        for (u32 j = 0; j < descs[e].dim; ++j) {              // an extension column = three base-field columns
            memcpy(&B.d[n], descs + e, sizeof(EvalDescK));
            B.d[n].offset += j; B.d[n].dim = 1; B.out[n] = slot++;
            if (++n == (u32)EV_LANES) flush();
        }
    }
    flush();
    ZK_HIP(hipGetLastError());
    for (u32 e0 = 0; e0 < n_ev; e0 += 64) {
        EvalFinalBatch F; memset(&F, 0, sizeof F);
        const u32 m = std::min<u32>(64, n_ev - e0);
        for (u32 i = 0; i < m; ++i) { F.slot[i] = slot0[e0 + i]; F.dim[i] = descs[e0 + i].dim; }
        hipLaunchKernelGGL(evals_final_kernel, dim3(m), dim3(64), 0, st, (const u64*)partial.u(), F, e0, n_row_blocks, d_out);
    }
    ZK_HIP(hipGetLastError());
}
// the reference's form: every section extended, rows k 2^ext, LEv / LpEv built with shift 49
void evals_dev(const EvalDescHost* descs, uint32_t n_ev, uint32_t nbits, uint32_t ext, const u64* d_LEv, const u64* d_LpEv,
               u64* d_out, hipStream_t st) {
    std::vector<EvalDescKHost> k(n_ev);
    for (u32 e = 0; e < n_ev; ++e) k[e] = EvalDescKHost{descs[e].buf, descs[e].prime ? d_LpEv : d_LEv, descs[e].width, descs[e].offset, descs[e].dim, ext};
    evals_k_dev(k.data(), n_ev, nbits, d_out, st);
}


// ---- calculate_H1H2 (stark_gen.rs:624-651) on the device ------------------------------------------------------------------
// The reference maps every table row t[j] to its LAST index j (a hash map filled in order), looks every f[i] up, and sorts
// the 2n pairs (index, value) stably by index with the table rows first: the sorted list is t[j] repeated 1 + c_j times for
// j = 0, 1, ..., where c_j counts the f's that found index j; h1 and h2 are its even and odd entries.  Here:
//   1. an open-addressing table over the indices of t (4n slots, linear probing; equal keys keep the larger index by atomicMax),
//   2. one lookup per f[i] -> j, a histogram c_j (and the smallest i that finds nothing: the reference's error),
//   3. an exclusive scan of 1 + c_j -> where index j starts in the sorted list,
//   4. output position q belongs to the last j whose start is <= q (binary search): h1[i] = t[owner(2i)], h2[i] = t[owner(2i + 1)].
namespace {
__device__ __forceinline__ u64 h1h2_hash(const u64* __restrict__ e) {
    u64 h = e[0] * 0x9E3779B97F4A7C15ull ^ (e[1] + 0x7F4A7C15ull) * 0xD1B54A32D192ED03ull ^ e[2] * 0x2545F4914F6CDD1Dull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}
__device__ __forceinline__ bool h1h2_eq(const u64* __restrict__ a, const u64* __restrict__ b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

__global__ __launch_bounds__(256) void h1h2_insert_kernel(const u64* __restrict__ t, u64 n, u64* __restrict__ slots /* idx + 1, 0 = empty */, u64 mask) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64* e = t + 3 * i;
    for (u64 s = h1h2_hash(e) & mask;; s = (s + 1) & mask) {
        const u64 cur = atomicCAS((unsigned long long*)&slots[s], 0ull, (unsigned long long)(i + 1));
        if (cur == 0) return;                                             // claimed an empty slot
        if (h1h2_eq(t + 3 * (cur - 1), e)) { atomicMax((unsigned long long*)&slots[s], (unsigned long long)(i + 1)); return; }   // same value: the last index wins
    }
}
__global__ __launch_bounds__(256) void h1h2_lookup_kernel(const u64* __restrict__ f, const u64* __restrict__ t, u64 n, const u64* __restrict__ slots,
                                                          u64 mask, u64* __restrict__ count /* [n], starts at 1 */, u64* __restrict__ missing) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64* e = f + 3 * i;
    for (u64 s = h1h2_hash(e) & mask;; s = (s + 1) & mask) {
        const u64 cur = slots[s];
        if (cur == 0) { atomicMin((unsigned long long*)missing, (unsigned long long)i); return; }
        if (h1h2_eq(t + 3 * (cur - 1), e)) { atomicAdd((unsigned long long*)&count[cur - 1], 1ull); return; }
    }
}
__global__ __launch_bounds__(256) void h1h2_fill_kernel(u64* __restrict__ p, u64 n, u64 v) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// exclusive scan of u64 counters, 4096 per block: block sums, a one-block scan of those (<= 2^16 of them for n <= 2^28), add back
constexpr int SCAN_PER = 16;
__global__ __launch_bounds__(256) void scan_block_kernel(const u64* __restrict__ in, u64 n, u64* __restrict__ out, u64* __restrict__ sums) {
    __shared__ u64 sh[256];
    const u64 base = ((u64)blockIdx.x * 256 + threadIdx.x) * SCAN_PER;
    u64 v[SCAN_PER], tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) { v[k] = base + k < n ? in[base + k] : 0; tot += v[k]; }
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const u64 add = threadIdx.x >= (u32)d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    u64 run = sh[threadIdx.x] - tot;                                      // exclusive prefix of this lane inside the block
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) sums[blockIdx.x] = sh[255];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(u64* __restrict__ sums, u64 nb) {   // one block, in place, exclusive
    __shared__ u64 sh[1024];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nb; base += 1024) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < nb ? sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const u64 add = threadIdx.x >= (u32)d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void scan_add_kernel(u64* __restrict__ out, u64 n, const u64* __restrict__ sums) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += sums[i / (256 * SCAN_PER)];
}
__global__ __launch_bounds__(256) void h1h2_emit_kernel(const u64* __restrict__ t, const u64* __restrict__ start /* [n] exclusive scan */, u64 n,
                                                        u64* __restrict__ h1, u64* __restrict__ h2) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;          // position in the sorted list of 2n values
    if (q >= 2 * n) return;
    u64 lo = 0, hi = n;                                                   // last j with start[j] <= q
    while (hi - lo > 1) { const u64 mid = (lo + hi) >> 1; if (start[mid] <= q) lo = mid; else hi = mid; }
    u64* __restrict__ o = (q & 1 ? h2 : h1) + 3 * (q >> 1);
    o[0] = t[3 * lo]; o[1] = t[3 * lo + 1]; o[2] = t[3 * lo + 2];
}
}  // namespace

// d_work: 4*pow2ceil(n) + 2n + n/4096 + 8 words.  *d_missing (a word of d_work, see h1h2_missing_word) ends as the smallest i
// whose f[i] is not in t, or ~0.
uint64_t h1h2_work_words(uint64_t n) {
    u64 m = 4; while (m < 4 * n) m <<= 1;
    return m + 2 * n + (n + 4095) / 4096 + 8;
}
void calculate_h1h2_dev(const u64* d_f, const u64* d_t, uint64_t n, u64* d_h1, u64* d_h2, u64* d_work, u64** d_missing, hipStream_t st) {
    u64 m = 4; while (m < 4 * n) m <<= 1;
    u64* slots = d_work; u64* count = slots + m; u64* start = count + n; u64* sums = start + n; u64* missing = sums + (n + 4095) / 4096 + 1;
    *d_missing = missing;
    const u64 nsb = (n + 4095) / 4096;
    ZK_HIP(hipMemsetAsync(slots, 0, m * 8, st));
    ZK_HIP(hipMemsetAsync(missing, 0xFF, 8, st));
    hipLaunchKernelGGL(h1h2_fill_kernel, grid1(n), dim3(256), 0, st, count, n, (u64)1);
    hipLaunchKernelGGL(h1h2_insert_kernel, grid1(n), dim3(256), 0, st, d_t, n, slots, m - 1);
    hipLaunchKernelGGL(h1h2_lookup_kernel, grid1(n), dim3(256), 0, st, d_f, d_t, n, slots, m - 1, count, missing);
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nsb), dim3(256), 0, st, count, n, start, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nsb);
    hipLaunchKernelGGL(scan_add_kernel, grid1(n), dim3(256), 0, st, start, n, sums);
    hipLaunchKernelGGL(h1h2_emit_kernel, grid1(2 * n), dim3(256), 0, st, d_t, start, n, d_h1, d_h2);
    ZK_HIP(hipGetLastError());
}

void qsplit_dev(const u64* d_qq1, uint32_t nbits, uint32_t q_dim, uint32_t q_deg, u64* d_qq2, hipStream_t st) {
    const u64 N = 1ull << nbits;
    const u64 shift_inv_n = gl::hpow(gl::hinv(49), N);
    hipLaunchKernelGGL(qsplit_kernel, grid1(N * q_deg * q_dim), dim3(256), 0, st, d_qq1, nbits, q_dim, q_deg, shift_inv_n, d_qq2);
    ZK_HIP(hipGetLastError());
}

}  // namespace zk
