// Curve-generic body of the G1 MSM (see msm.hip): included once per curve inside that curve's namespace,
// which provides NL (32-bit limbs of Fq), FQ_Q(i), FQ_ONE(i) (R mod q), FQ_INV (-q^-1 mod 2^32) and the
// generator in Montgomery form GEN_X(i), GEN_Y(i).  The curve is y^2 = x^3 + b with a = 0 (the
// formulas below never touch b).  No include guard on purpose.
struct fq { u32 l[NL]; };

__device__ __forceinline__ bool fq_is_zero(const fq& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) o |= a.l[i];
    return o == 0;
}
__device__ __forceinline__ bool fq_eq(const fq& a, const fq& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
// r = a - q if a >= q (a < 2q, optional incoming carry bit)
__device__ __forceinline__ fq fq_reduce_once(const fq& a, u32 carry) {
    fq t; long long br = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) { long long d = (long long)a.l[i] - FQ_Q(i) + br; t.l[i] = (u32)d; br = d >> 32; }
    const bool ge = carry || br == 0;  // no final borrow <=> a >= q
    fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = ge ? t.l[i] : a.l[i];
    return r;
}
__device__ __forceinline__ fq fq_add(const fq& a, const fq& b) {
    fq s; u64 c = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) { c += (u64)a.l[i] + b.l[i]; s.l[i] = (u32)c; c >>= 32; }
    return fq_reduce_once(s, (u32)c);
}
__device__ __forceinline__ fq fq_sub(const fq& a, const fq& b) {
    fq d; long long br = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) { long long x = (long long)a.l[i] - b.l[i] + br; d.l[i] = (u32)x; br = x >> 32; }
    const bool neg = br != 0;
    u64 c = 0; fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i) { c += (u64)d.l[i] + (neg ? FQ_Q(i) : 0u); r.l[i] = (u32)c; c >>= 32; }
    return r;
}
__device__ __forceinline__ fq fq_dbl(const fq& a) { return fq_add(a, a); }
// Montgomery product a*b*R^-1 mod q (CIOS, 32-bit limbs)
#ifndef FQ_MUL_ATTR
#define FQ_MUL_ATTR __forceinline__
#endif
__device__ FQ_MUL_ATTR fq fq_mul(const fq& a, const fq& b) {
    u32 t[NL + 2];
#pragma unroll
    for (int i = 0; i < NL + 2; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) { c += (u64)a.l[j] * b.l[i] + t[j]; t[j] = (u32)c; c >>= 32; }
        c += t[NL]; t[NL] = (u32)c; t[NL + 1] = (u32)(c >> 32);
        const u32 m = t[0] * FQ_INV;
        c = ((u64)m * FQ_Q(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < NL; ++j) { c += (u64)m * FQ_Q(j) + t[j]; t[j - 1] = (u32)c; c >>= 32; }
        c += t[NL]; t[NL - 1] = (u32)c; t[NL] = t[NL + 1] + (u32)(c >> 32);
    }
    fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = t[i];
    return fq_reduce_once(r, t[NL]);
}
__device__ __forceinline__ fq fq_sqr(const fq& a) { return fq_mul(a, a); }
__device__ fq fq_inv(const fq& a) {  // a^(q-2)
    fq r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = FQ_ONE(i);
    for (int bit = 32 * NL - 1; bit >= 0; --bit) {
        r = fq_sqr(r);
        u32 w = FQ_Q(bit >> 5);  // exponent q - 2: only limb 0 differs
        if ((bit >> 5) == 0) w -= 2;
        // FQ_Q with a runtime index is a constant-array lookup; fine off the hot path
        if ((w >> (bit & 31)) & 1) r = fq_mul(r, a);
    }
    return r;
}

// XYZZ coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0
struct xyzz { fq X, Y, ZZ, ZZZ; };
struct aff { fq x, y; };

__device__ __forceinline__ xyzz pt_inf() {
    xyzz p;
#pragma unroll
    for (int i = 0; i < NL; ++i) { p.X.l[i] = 0; p.Y.l[i] = 0; p.ZZ.l[i] = 0; p.ZZZ.l[i] = 0; }
    return p;
}
__device__ __forceinline__ bool pt_is_inf(const xyzz& p) { return fq_is_zero(p.ZZ); }
__device__ xyzz pt_dbl_aff(const aff& a) {  // mdbl-2008-s-1 (a = 0)
    fq U = fq_dbl(a.y), V = fq_sqr(U), W = fq_mul(U, V), S = fq_mul(a.x, V);
    fq xx = fq_sqr(a.x), M = fq_add(fq_dbl(xx), xx);
    xyzz r;
    r.X = fq_sub(fq_sqr(M), fq_dbl(S));
    r.Y = fq_sub(fq_mul(M, fq_sub(S, r.X)), fq_mul(W, a.y));
    r.ZZ = V; r.ZZZ = W;
    return r;
}
__device__ xyzz pt_dbl(const xyzz& p) {  // dbl-2008-s-1 (a = 0)
    if (pt_is_inf(p)) return p;
    fq U = fq_dbl(p.Y), V = fq_sqr(U), W = fq_mul(U, V), S = fq_mul(p.X, V);
    fq xx = fq_sqr(p.X), M = fq_add(fq_dbl(xx), xx);
    xyzz r;
    r.X = fq_sub(fq_sqr(M), fq_dbl(S));
    r.Y = fq_sub(fq_mul(M, fq_sub(S, r.X)), fq_mul(W, p.Y));
    r.ZZ = fq_mul(V, p.ZZ); r.ZZZ = fq_mul(W, p.ZZZ);
    return r;
}
__device__ xyzz pt_madd(const xyzz& p, const aff& a) {  // madd-2008-s
    if (pt_is_inf(p)) {
        xyzz r; r.X = a.x; r.Y = a.y;
#pragma unroll
        for (int i = 0; i < NL; ++i) { r.ZZ.l[i] = FQ_ONE(i); r.ZZZ.l[i] = FQ_ONE(i); }
        return r;
    }
    fq U2 = fq_mul(a.x, p.ZZ), S2 = fq_mul(a.y, p.ZZZ);
    fq Pd = fq_sub(U2, p.X), Rd = fq_sub(S2, p.Y);
    if (fq_is_zero(Pd)) return fq_is_zero(Rd) ? pt_dbl_aff(a) : pt_inf();
    fq PP = fq_sqr(Pd), PPP = fq_mul(Pd, PP), Qv = fq_mul(p.X, PP);
    xyzz r;
    r.X = fq_sub(fq_sub(fq_sqr(Rd), PPP), fq_dbl(Qv));
    r.Y = fq_sub(fq_mul(Rd, fq_sub(Qv, r.X)), fq_mul(p.Y, PPP));
    r.ZZ = fq_mul(p.ZZ, PP); r.ZZZ = fq_mul(p.ZZZ, PPP);
    return r;
}
__device__ xyzz pt_add(const xyzz& p, const xyzz& q) {  // add-2008-s
    if (pt_is_inf(p)) return q;
    if (pt_is_inf(q)) return p;
    fq U1 = fq_mul(p.X, q.ZZ), U2 = fq_mul(q.X, p.ZZ), S1 = fq_mul(p.Y, q.ZZZ), S2 = fq_mul(q.Y, p.ZZZ);
    fq Pd = fq_sub(U2, U1), Rd = fq_sub(S2, S1);
    if (fq_is_zero(Pd)) return fq_is_zero(Rd) ? pt_dbl(p) : pt_inf();
    fq PP = fq_sqr(Pd), PPP = fq_mul(Pd, PP), Qv = fq_mul(U1, PP);
    xyzz r;
    r.X = fq_sub(fq_sub(fq_sqr(Rd), PPP), fq_dbl(Qv));
    r.Y = fq_sub(fq_mul(Rd, fq_sub(Qv, r.X)), fq_mul(S1, PPP));
    r.ZZ = fq_mul(fq_mul(p.ZZ, q.ZZ), PP); r.ZZZ = fq_mul(fq_mul(p.ZZZ, q.ZZZ), PPP);
    return r;
}
__device__ __forceinline__ xyzz pt_neg(const xyzz& p) {
    xyzz r = p;
    fq z;
#pragma unroll
    for (int i = 0; i < NL; ++i) z.l[i] = 0;
    if (!fq_is_zero(p.Y)) r.Y = fq_sub(z, p.Y);
    return r;
}

constexpr int C_BITS = 16, N_WIN = 16, N_BUCKET = 1 << C_BITS;  // 254-bit scalars: 16 windows of 16 bits

__global__ void msm_count_kernel(const u32* __restrict__ scalars, u64 n, u32* __restrict__ counts) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;  // one lane per (point, window)
    if (t >= n * N_WIN) return;
    const u64 i = t / N_WIN; const u32 w = t % N_WIN;
    const u32 word = scalars[i * 8 + (w >> 1)];
    const u32 d = (w & 1) ? word >> 16 : word & 0xFFFF;
    if (d) atomicAdd(&counts[w * N_BUCKET + d], 1u);
}
__global__ void msm_scatter_kernel(const u32* __restrict__ scalars, u64 n, const u32* __restrict__ offsets,
                                   u32* __restrict__ cursors, u32* __restrict__ idx) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N_WIN) return;
    const u64 i = t / N_WIN; const u32 w = t % N_WIN;
    const u32 word = scalars[i * 8 + (w >> 1)];
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
__global__ void scan_tops_kernel(u32* __restrict__ block_sum, u32 nb) {  // nb <= 1024: one lane, serial
    if (threadIdx.x | blockIdx.x) return;
    u32 acc = 0;
    for (u32 b = 0; b < nb; ++b) { u32 t = block_sum[b]; block_sum[b] = acc; acc += t; }
}
__global__ void scan_add_kernel(u32* __restrict__ out, const u32* __restrict__ block_sum) {
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 0] += block_sum[blockIdx.x];
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 1] += block_sum[blockIdx.x];
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 2] += block_sum[blockIdx.x];
    out[blockIdx.x * 1024 + threadIdx.x * 4 + 3] += block_sum[blockIdx.x];
}

__device__ __forceinline__ aff load_aff(const u32* __restrict__ bases, u32 i) {
    aff a;
    const uint4* p = (const uint4*)(bases + (u64)i * (2 * NL));  // 64 B (BN254) or 96 B (BLS12-381) per point, 16-byte aligned
#pragma unroll
    for (int k = 0; k < NL / 4; ++k) {
        const uint4 vx = p[k], vy = p[NL / 4 + k];
        a.x.l[4 * k] = vx.x; a.x.l[4 * k + 1] = vx.y; a.x.l[4 * k + 2] = vx.z; a.x.l[4 * k + 3] = vx.w;
        a.y.l[4 * k] = vy.x; a.y.l[4 * k + 1] = vy.y; a.y.l[4 * k + 2] = vy.z; a.y.l[4 * k + 3] = vy.w;
    }
    return a;
}
__global__ __launch_bounds__(64) void msm_accumulate_kernel(const u32* __restrict__ bases, const u32* __restrict__ offsets,
                                                            const u32* __restrict__ counts, const u32* __restrict__ idx,
                                                            xyzz* __restrict__ buckets) {
    const u32 key = blockIdx.x * blockDim.x + threadIdx.x;  // window * 2^16 + digit
    xyzz acc = pt_inf();
    const u32 n = counts[key], off = offsets[key];
    for (u32 k = 0; k < n; ++k) acc = pt_madd(acc, load_aff(bases, idx[off + k]));
    buckets[key] = acc;
}
// One level of the radix-16 hierarchy that computes sum_k k*B_k per window.  An item (S, A) stands
// for a block of m = 16^level consecutive buckets: S = their sum, A = sum (local index) * bucket.
// 16 neighbouring blocks combine as S' = sum_j S_j, A' = sum_j A_j + m * sum_j j*S_j (running-sum
// trick for the last term).  Level 0 reads the buckets themselves (A = 0).  After 4 levels the one
// item left per window holds A = sum_k k*B_k.  Every level keeps 1/16 of the lanes of the one
// before: 2^16, 2^12, 2^8, 2^4 -- the serial chain per lane is 47 additions, not 65536.
__global__ __launch_bounds__(64) void msm_reduce_level_kernel(const xyzz* __restrict__ S_in, const xyzz* __restrict__ A_in,
                                                              xyzz* __restrict__ S_out, xyzz* __restrict__ A_out,
                                                              u32 n_out, int level) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_out) return;
    const xyzz* s = S_in + (u64)g * 16;
    xyzz run = pt_inf(), acc = pt_inf();
    for (int j = 15; j >= 1; --j) { run = pt_add(run, s[j]); acc = pt_add(acc, run); }
    run = pt_add(run, s[0]);
    if (level > 0) {
        for (int k = 0; k < 4 * level; ++k) acc = pt_dbl(acc);
        const xyzz* a = A_in + (u64)g * 16;
        for (int j = 0; j < 16; ++j) acc = pt_add(acc, a[j]);
    }
    S_out[g] = run; A_out[g] = acc;
}
__global__ void msm_final_kernel(const xyzz* __restrict__ win, u32* __restrict__ out /* 2*NL words + flag */) {
    if (threadIdx.x | blockIdx.x) return;
    xyzz acc = pt_inf();
    for (int w = N_WIN - 1; w >= 0; --w) {
        for (int k = 0; k < C_BITS; ++k) acc = pt_dbl(acc);
        acc = pt_add(acc, win[w]);
    }
    if (pt_is_inf(acc)) { for (int i = 0; i < 2 * NL; ++i) out[i] = 0; out[2 * NL] = 1; return; }
    // x = X/ZZ, y = Y/ZZZ ; 1/ZZ = (ZZ * 1/ZZZ)^2 because ZZ^3 = ZZZ^2
    fq izzz = fq_inv(acc.ZZZ), t = fq_mul(acc.ZZ, izzz), izz = fq_sqr(t);
    fq x = fq_mul(acc.X, izz), y = fq_mul(acc.Y, izzz);
    for (int i = 0; i < NL; ++i) { out[i] = x.l[i]; out[NL + i] = y.l[i]; }
    out[2 * NL] = 0;
}

// synthetic bases for benches/tests: P_i = [k_i]G, G = the curve's generator; k_i 64-bit, non-zero
__global__ __launch_bounds__(64) void g1_mul_generator_kernel(const u64* __restrict__ k, u64 n, u32* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    aff g;
    for (int j = 0; j < NL; ++j) { g.x.l[j] = GEN_X(j); g.y.l[j] = GEN_Y(j); }
    const u64 e = k[i];
    xyzz acc = pt_inf();
    for (int b = 63; b >= 0; --b) {
        acc = pt_dbl(acc);
        if ((e >> b) & 1) acc = pt_madd(acc, g);
    }
    u32* o = out + i * (2 * NL);
    if (pt_is_inf(acc)) { for (int j = 0; j < 2 * NL; ++j) o[j] = 0; return; }
    fq izzz = fq_inv(acc.ZZZ), t = fq_mul(acc.ZZ, izzz), izz = fq_sqr(t);
    fq x = fq_mul(acc.X, izz), y = fq_mul(acc.Y, izzz);
    for (int j = 0; j < NL; ++j) { o[j] = x.l[j]; o[NL + j] = y.l[j]; }
}


void g1_mul_generator_dev(const u64* d_k, uint64_t n, void* d_bases, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(g1_mul_generator_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_k, n, (u32*)d_bases);
    ZK_HIP(hipGetLastError());
}

// d_out: 2*NL + 1 u32 words (x, y Montgomery, infinity flag)
void msm_g1_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, hipStream_t st) {
    ZK_REQUIRE(n >= 1 && n < (1ull << 28), "msm: n out of range");
    const size_t n_keys = (size_t)N_WIN * N_BUCKET;
    DevBuf counts, offsets, cursors, tops, idx, buckets, S0, A0, S1, A1;
    counts.reserve(n_keys * 4); offsets.reserve(n_keys * 4); cursors.reserve(n_keys * 4); tops.reserve(1024 * 4);
    idx.reserve((size_t)n * N_WIN * 4);
    buckets.reserve(n_keys * sizeof(xyzz));
    S0.reserve(n_keys / 16 * sizeof(xyzz)); A0.reserve(n_keys / 16 * sizeof(xyzz));
    S1.reserve(n_keys / 256 * sizeof(xyzz)); A1.reserve(n_keys / 256 * sizeof(xyzz));
    ZK_HIP(hipMemsetAsync(counts.p, 0, n_keys * 4, st));
    ZK_HIP(hipMemsetAsync(cursors.p, 0, n_keys * 4, st));
    const u64 total = n * N_WIN;
    const unsigned gb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(msm_count_kernel, dim3(gb), dim3(256), 0, st, (const u32*)d_scalars, n, (u32*)counts.p);
    ZK_HIP(hipGetLastError());
    const unsigned nb = (unsigned)(n_keys / 1024);
    hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(256), 0, st, (const u32*)counts.p, (u32*)offsets.p, (u32*)tops.p);
    hipLaunchKernelGGL(scan_tops_kernel, dim3(1), dim3(64), 0, st, (u32*)tops.p, nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, st, (u32*)offsets.p, (const u32*)tops.p);
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(gb), dim3(256), 0, st, (const u32*)d_scalars, n, (const u32*)offsets.p,
                       (u32*)cursors.p, (u32*)idx.p);
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3((unsigned)(n_keys / 64)), dim3(64), 0, st, (const u32*)d_bases,
                       (const u32*)offsets.p, (const u32*)counts.p, (const u32*)idx.p, (xyzz*)buckets.p);
    ZK_HIP(hipGetLastError());
    // radix-16 reduction hierarchy: ping-pong (S, A) arrays of n_keys/16 items
    const xyzz* s_in = (const xyzz*)buckets.p; const xyzz* a_in = nullptr;
    u32 n_out = (u32)(n_keys / 16);
    for (int level = 0; level < C_BITS / 4; ++level, n_out /= 16) {
        xyzz* s_out = (xyzz*)(level & 1 ? S1.p : S0.p); xyzz* a_out = (xyzz*)(level & 1 ? A1.p : A0.p);
        hipLaunchKernelGGL(msm_reduce_level_kernel, dim3((n_out + 63) / 64), dim3(64), 0, st, s_in, a_in, s_out, a_out, n_out, level);
        s_in = s_out; a_in = a_out;
    }
    ZK_HIP(hipGetLastError());
    hipLaunchKernelGGL(msm_final_kernel, dim3(1), dim3(64), 0, st, a_in, (u32*)d_out);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(st));  // the pooled scratch above is released at scope exit
}

