// Scalar-field transforms and the pointwise work of the Groth16 quotient on gfx950, generic over the field.
// Included inside a namespace that has already pulled in fr29_consts.hip.h + fe29_impl.hip.h and defines
// FRN_S (2-adicity), FRN_ROOT (8 x u32: a primitive 2^S-th root of unity, Montgomery R = 2^256; 7^((r-1)/2^S), the
// value ff's derive macro gives bellman's Fr::root_of_unity()).  No include guard on purpose.
//
// What this replaces: bellman_ce's EvaluationDomain::{fft, ifft, coset_fft, icoset_fft, mul_assign, sub_assign,
// divide_by_z_on_coset} as driven by groth16/prover.rs create_proof (third-party; call site
// groth16/src/groth16.rs:93).  Natural order in and out, omega = ROOT^(2^(S-k)), coset generator 7.
//
// Layout: a vector of n field elements is stored limb-major, u32 v[9][n] (29-bit limbs, Montgomery R' = 2^261), so
// that a wave's access to one limb of 64 consecutive elements is one 256-byte line.  A transform is
// ceil(k/3) out-of-place Stockham passes of radix <= 8; a lane owns the 8 points {j + t n/8}, multiplies them by
// the pass twiddles w^(k t) from the n-entry table, runs a decimation-in-frequency 8-point transform in
// registers and scatters to (j - k) 8 + k + bitrev(t) L.  Integer-ALU bound: 13 field products per 8 points per
// pass (225 instructions each, fe29_impl.hip.h) against 72 bytes of traffic per point.
// Value bounds: stored values are < 16q with normalised limbs; every element meets a product (twiddle, table or
// R' mod q) on load, which brings it below 2q; three butterfly levels then grow sums to < 16q and differences
// (biased by 2q, 4q, 8q) likewise; products inside the levels take at most 8q x 2q (A*B = 16 <= 68).

__device__ __forceinline__ fe soa_load(const u32* __restrict__ base, u64 n, u64 i) {
    fe r;
#pragma unroll
    for (int l = 0; l < NR; ++l) r.l[l] = base[(u64)l * n + i];
    return r;
}
__device__ __forceinline__ void soa_store(u32* __restrict__ base, u64 n, u64 i, const fe& v) {
#pragma unroll
    for (int l = 0; l < NR; ++l) base[(u64)l * n + i] = v.l[l];
}
__device__ __forceinline__ fe fe_small(u32 v) {            // the plain value v (v < 2^29) in Montgomery form
    fe x = fe_zero(); x.l[0] = v;
    fe c;
#pragma unroll
    for (int i = 0; i < NR; ++i) c.l[i] = RRP29(i);
    return fe_mul(x, c);
}
__device__ __forceinline__ fe fe_renorm(const fe& a) { return fe_mul(a, fe_one()); }   // any value < 68q -> < 2q

// per-domain scalars and the square chains the table kernels multiply together
struct FrDomainConsts {
    fe w2[32], g2[32], gi2[32];   // w^(2^b), 7^(2^b), 7^-(2^b)
    fe minv, zinv;                // 1/n, 1/(7^n - 1)
};
__global__ void frn_setup_kernel(FrDomainConsts* dc, int logn) {
    if (threadIdx.x || blockIdx.x) return;
    u32 rw[NL] = {FRN_ROOT};
    fe w = fe_from_std(rw);
    for (int i = logn; i < FRN_S; ++i) w = fe_sqr(w);
    fe g = fe_small(7), gi = fe_inv(g);
    fe gn = g;                                            // 7^n after logn squarings
    for (int b = 0; b < 32; ++b) {
        dc->w2[b] = w; dc->g2[b] = g; dc->gi2[b] = gi;
        if (b < logn) gn = fe_sqr(gn);
        w = fe_sqr(w); g = fe_sqr(g); gi = fe_sqr(gi);
    }
    fe two = fe_small(2), nn = fe_one();
    for (int b = 0; b < logn; ++b) nn = fe_mul(nn, two);
    dc->minv = fe_inv(nn);
    dc->zinv = fe_inv(fe_sub<2>(gn, fe_one()));
}
// out[i] = scale * prod_{bit b of i} chain[b]
__global__ __launch_bounds__(256) void frn_pow_table_kernel(const fe* __restrict__ chain, const fe* __restrict__ scale, u32* __restrict__ out, u64 n) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    fe acc = scale ? *scale : fe_one();
    for (int b = 0; (i >> b) != 0; ++b)
        if ((i >> b) & 1) acc = fe_mul(acc, chain[b]);
    soa_store(out, n, i, acc);
}

// compile-time loops: with `#pragma unroll` the 8-element array x[] stayed in the private segment (304 B of scratch per
// lane, passes three times slower)
template <int B, int E, class F>
__device__ __forceinline__ void frn_static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); frn_static_for<B + 1, E>(f); }
}
struct FrnBatch { const u32* in[3]; u32* out[3]; };   // up to three vectors per launch (blockIdx.y): the a, b, c of the quotient
template <int LOGR>
__global__ __launch_bounds__(256) void frn_pass_kernel(const FrnBatch bt, const u32* __restrict__ W, int logn, int logL, int inverse,
                                                       const u32* __restrict__ pre, const u32* __restrict__ post, u64 post_n) {
    constexpr int R = 1 << LOGR;
    const u32* __restrict__ in = bt.in[blockIdx.y];
    u32* __restrict__ out = bt.out[blockIdx.y];
    const u64 N = 1ull << logn, M = N >> LOGR;
    const u64 j = blockIdx.x * 256ull + threadIdx.x;
    if (j >= M) return;
    const u64 k = j & ((1ull << logL) - 1);
    fe x[R];
    frn_static_for<0, R>([&](auto T) {
        constexpr int t = decltype(T)::value;
        const u64 idx = j + (u64)t * M;
        x[t] = soa_load(in, N, idx);
        if (pre) x[t] = fe_mul(x[t], soa_load(pre, N, idx));
    });
    if (logL > 0) {
        if (!pre) x[0] = fe_renorm(x[0]);
        frn_static_for<1, R>([&](auto T) {
            constexpr int t = decltype(T)::value;
            u64 e = (k * (u64)t) << (logn - logL - LOGR);
            if (inverse) e = (N - e) & (N - 1);
            x[t] = fe_mul(x[t], soa_load(W, N, e));
        });
    }
    // decimation in frequency: natural order in, bit-reversed order out
    frn_static_for<0, LOGR>([&](auto SI) {
        constexpr int s = LOGR - 1 - decltype(SI)::value, half = 1 << s;
        frn_static_for<0, R>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if constexpr ((u & half) == 0) {
                const fe a = x[u], b = x[u + half];
                x[u] = fe_add(a, b);
                fe d = fe_sub<(2 << (LOGR - 1 - s))>(a, b);       // biases 2q, 4q, 8q for inputs < 2q, 4q, 8q
                constexpr int p = u & (half - 1);
                if constexpr (p != 0) {
                    u64 e = (N >> (s + 1)) * (u64)p;
                    if (inverse) e = N - e;
                    d = fe_mul(d, soa_load(W, N, e));              // wave-uniform address
                }
                x[u + half] = d;
            }
        });
    });
    const u64 obase = ((j - k) << LOGR) + k;
    frn_static_for<0, R>([&](auto T) {
        constexpr int t = decltype(T)::value;
        int tt = 0;
        for (int b = 0; b < LOGR; ++b) tt |= ((t >> b) & 1) << (LOGR - 1 - b);
        const u64 o = obase + ((u64)tt << logL);
        fe v = x[t];
        if (post) v = fe_mul(v, soa_load(post, post_n, post_n == 1 ? 0 : o));
        soa_store(out, N, o, v);
    });
}
// n == 1 and other degenerate shapes: v[i] *= pre[i] * post[i]
__global__ __launch_bounds__(256) void frn_scale_kernel(u32* __restrict__ v, u64 n, const u32* __restrict__ pre, const u32* __restrict__ post, u64 post_n) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    fe x = fe_renorm(soa_load(v, n, i));
    if (pre) x = fe_mul(x, soa_load(pre, n, i));
    if (post) x = fe_mul(x, soa_load(post, post_n, post_n == 1 ? 0 : i));
    soa_store(v, n, i, x);
}

// ---- layout changes at the boundary ----------------------------------------------------------------------------
// n_valid elements of 8 x u32 (Montgomery R = 2^256, bellman's Fr) -> limb-major internal form, zero-padded to n
__global__ __launch_bounds__(256) void frn_from_std_kernel(const u32* __restrict__ in, u32* __restrict__ out, u64 n, u64 n_valid) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    fe x = fe_zero();
    if (i < n_valid) {
        u32 w[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) w[k] = in[i * NL + k];
        x = fe_from_std(w);
    }
    soa_store(out, n, i, x);
}
__global__ __launch_bounds__(256) void frn_to_std_kernel(const u32* __restrict__ in, u32* __restrict__ out, u64 n, u64 n_out) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n_out) return;
    u32 w[NL];
    fe_to_std(soa_load(in, n, i), w);
#pragma unroll
    for (int k = 0; k < NL; ++k) out[i * NL + k] = w[k];
}
// internal form -> canonical integers (FrRepr: what the multi-scalar sums take)
__global__ __launch_bounds__(256) void frn_to_canon_kernel(const u32* __restrict__ in, u32* __restrict__ out, u64 n, u64 n_out) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n_out) return;
    fe one = fe_zero(); one.l[0] = 1;
    const fe x = fe_canon(fe_mul(soa_load(in, n, i), one));
    u32 w[NL];
#pragma unroll
    for (int jj = 0; jj < NL; ++jj) {
        const int bit = 32 * jj, k = bit / LB, s = bit % LB;
        u32 v = x.l[k] >> s;
        if (k + 1 < NR) v |= x.l[k + 1] << (LB - s);
        if (k + 2 < NR && 2 * LB - s < 32) v |= x.l[k + 2] << (2 * LB - s);
        w[jj] = v;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) out[i * NL + k] = w[k];
}
// canonical integers (8 x u32 each, < r) -> internal form, element-major (9 x u32 each): the witness
__global__ __launch_bounds__(256) void frn_canon_to_fe_kernel(const u32* __restrict__ in, u32* __restrict__ out, u64 n) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    u32 w[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) w[k] = in[i * NL + k];
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
    for (int k = 0; k < NR; ++k) c.l[k] = RRP29(k);
    x = fe_mul(x, c);
#pragma unroll
    for (int k = 0; k < NR; ++k) out[i * NR + k] = x.l[k];
}

// ---- the quotient's pointwise step: a <- (a b - c) / Z on the coset (mul_assign, sub_assign, divide_by_z_on_coset)
__global__ __launch_bounds__(256) void frn_quotient_pointwise_kernel(u32* __restrict__ a, const u32* __restrict__ b, const u32* __restrict__ c, const fe* __restrict__ zinv, u64 n) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const fe ab = fe_mul(soa_load(a, n, i), fe_renorm(soa_load(b, n, i)));             // 16q x 2q
    const fe d = fe_sub<2>(ab, fe_renorm(soa_load(c, n, i)));                          // < 4q
    soa_store(a, n, i, fe_mul(d, *zinv));                                              // < 2q: the next transform's input contract
}

// ---- ProvingAssignment::enforce's eval(): one lane per row of a CSR matrix, out[i] = sum coeff * w[col]; rows
// ---- beyond n_rows are the domain's zero padding
__global__ __launch_bounds__(256) void frn_r1cs_eval_kernel(const u64* __restrict__ row_ptr, const u32* __restrict__ cols, const u32* __restrict__ coeffs,
                                                            const u32* __restrict__ wit, u64 n_rows, u32* __restrict__ out, u64 n) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    fe acc = fe_zero();
    if (i < n_rows) {
        int pending = 0;
        for (u64 k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            fe cf, x;
            const u64 col = cols[k];
#pragma unroll
            for (int l = 0; l < NR; ++l) { cf.l[l] = coeffs[k * NR + l]; x.l[l] = wit[col * NR + l]; }
            acc = fe_add(acc, fe_mul(cf, x));
            if (++pending == 4) { acc = fe_renorm(acc); pending = 0; }   // < 2q + 4 * 2q between renormalisations
        }
        if (pending) acc = fe_renorm(acc);
    }
    soa_store(out, n, i, acc);
}
__device__ __forceinline__ void fe_store_canon(const fe& a /* Montgomery, < 68q */, u32* __restrict__ out) {
    fe one = fe_zero(); one.l[0] = 1;
    const fe x = fe_canon(fe_mul(a, one));
#pragma unroll
    for (int jj = 0; jj < NL; ++jj) {
        const int bit = 32 * jj, k = bit / LB, s = bit % LB;
        u32 v = x.l[k] >> s;
        if (k + 1 < NR) v |= x.l[k + 1] << (LB - s);
        if (k + 2 < NR && 2 * LB - s < 32) v |= x.l[k + 2] << (2 * LB - s);
        out[jj] = v;
    }
}
// density-indexed scalars (8 x u32 canonical each): out[i] = scale * w[idx[i]] (idx < 0: 0).  wit: canonical values,
// wit_fe: the same in internal form (element-major), used when a scale is given
__global__ __launch_bounds__(256) void frn_gather_kernel(const u32* __restrict__ wit, const u32* __restrict__ wit_fe, const int* __restrict__ idx, u64 n,
                                                         const fe* __restrict__ scale, u32* __restrict__ out) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const int s = idx[i];
    if (!scale || s < 0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) out[i * NL + k] = s >= 0 ? wit[(u64)s * NL + k] : 0u;
        return;
    }
    fe x;
#pragma unroll
    for (int l = 0; l < NR; ++l) x.l[l] = wit_fe[(u64)s * NR + l];
    fe_store_canon(fe_mul(x, *scale), out + i * NL);
}
// the blinding scalars r, s (canonical, rs[0..8) and rs[8..16)): their internal forms and the tail entries of the
// three scalar vectors: (r, 1), (s, 1), (r), (s r, s)
__global__ void frn_blind_kernel(const u32* __restrict__ rs, fe* __restrict__ rs_fe, u32* __restrict__ tail_a, u32* __restrict__ tail_b,
                                 u32* __restrict__ tail_cb, u32* __restrict__ tail_ca) {
    if (threadIdx.x || blockIdx.x) return;
    fe v[2];
    for (int t = 0; t < 2; ++t) {
        fe x;
        for (int k = 0; k < NR; ++k) {
            const int bit = LB * k, wi = bit >> 5, sh = bit & 31;
            u32 u = wi < NL ? rs[8 * t + wi] >> sh : 0;
            if (sh > 32 - LB && wi + 1 < NL) u |= rs[8 * t + wi + 1] << (32 - sh);
            x.l[k] = u & LMASK;
        }
        fe c;
        for (int k = 0; k < NR; ++k) c.l[k] = RRP29(k);
        v[t] = fe_mul(x, c);
        rs_fe[t] = v[t];
    }
    for (int k = 0; k < NL; ++k) {
        tail_a[k] = rs[k]; tail_a[NL + k] = k == 0;
        tail_b[k] = rs[8 + k]; tail_b[NL + k] = k == 0;
        tail_cb[k] = rs[k];
        tail_ca[NL + k] = rs[8 + k];
    }
    fe_store_canon(fe_mul(v[0], v[1]), tail_ca);
}

// ---- host side --------------------------------------------------------------------------------------------------
struct FrDomain {
    int logn = -1;
    DevBuf consts, W, GP, GIP;   // FrDomainConsts; w^i; 7^i; 7^-i / n   (limb-major, n entries each)
    const FrDomainConsts* dc() const { return (const FrDomainConsts*)consts.p; }
    const fe* minv() const { return &dc()->minv; }
    const fe* zinv() const { return &dc()->zinv; }
};
static std::mutex g_dom_mu;
static auto& g_domains = *new std::map<std::pair<int, int>, std::unique_ptr<FrDomain>>();   // (device, logn); never destroyed: the pool outlives it

static inline unsigned frn_blocks(u64 n) { return (unsigned)((n + 255) / 256); }

static const FrDomain& frn_domain(int logn, hipStream_t st) {
    if (logn < 0 || logn > FRN_S) throw std::runtime_error("fr ntt: domain of 2^" + std::to_string(logn) + " exceeds the field's 2-adicity");
    int dev = 0; ZK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_dom_mu);
    auto& slot = g_domains[{dev, logn}];
    if (slot) return *slot;
    auto d = std::make_unique<FrDomain>();
    d->logn = logn;
    const u64 n = 1ull << logn;
    d->consts.reserve(sizeof(FrDomainConsts));
    d->W.reserve(n * NR * 4); d->GP.reserve(n * NR * 4); d->GIP.reserve(n * NR * 4);
    auto* dc = (FrDomainConsts*)d->consts.p;
    hipLaunchKernelGGL(frn_setup_kernel, dim3(1), dim3(64), 0, st, dc, logn);
    hipLaunchKernelGGL(frn_pow_table_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const fe*)dc->w2, (const fe*)nullptr, (u32*)d->W.p, n);
    hipLaunchKernelGGL(frn_pow_table_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const fe*)dc->g2, (const fe*)nullptr, (u32*)d->GP.p, n);
    hipLaunchKernelGGL(frn_pow_table_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const fe*)dc->gi2, (const fe*)&dc->minv, (u32*)d->GIP.p, n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(st));
    slot = std::move(d);
    return *slot;
}

// one transform over limb-major buffers: src -> (ping-pong) -> returns the buffer holding the result (a or b).
// pre: n-entry table applied on load of the first pass; post: table (post_n = n) or one element (post_n = 1) on the
// stores of the last.  Input contract: values < 2q when pre == nullptr.
// nb <= 3 vectors at once: a[i] -> (ping-pong with b[i]) ; returns true when the results are in b[], false when in a[]
static bool frn_transform_batch(const FrDomain& D, u32* const* a, u32* const* b, int nb, bool inverse, const u32* pre, const u32* post, u64 post_n, hipStream_t st) {
    const int logn = D.logn;
    const u64 n = 1ull << logn;
    if (logn == 0) {
        for (int i = 0; i < nb; ++i) hipLaunchKernelGGL(frn_scale_kernel, dim3(1), dim3(256), 0, st, a[i], n, pre, post, post_n);
        ZK_HIP(hipGetLastError());
        return false;
    }
    int logL = 0;
    bool in_b = false;
    const int rem = logn % 3;
    const int n_pass = logn / 3 + (rem ? 1 : 0);
    for (int p = 0; p < n_pass; ++p) {
        const int lr = (p == 0 && rem) ? rem : 3;
        const u32* pr = p == 0 ? pre : nullptr;
        const u32* po = p == n_pass - 1 ? post : nullptr;
        const u64 threads = n >> lr;
        const dim3 grid(frn_blocks(threads), (unsigned)nb), blk(256);
        const u32* W = (const u32*)D.W.p;
        FrnBatch bt{};
        for (int i = 0; i < nb; ++i) { bt.in[i] = in_b ? b[i] : a[i]; bt.out[i] = in_b ? a[i] : b[i]; }
        if (lr == 1) hipLaunchKernelGGL(frn_pass_kernel<1>, grid, blk, 0, st, bt, W, logn, logL, (int)inverse, pr, po, post_n);
        else if (lr == 2) hipLaunchKernelGGL(frn_pass_kernel<2>, grid, blk, 0, st, bt, W, logn, logL, (int)inverse, pr, po, post_n);
        else hipLaunchKernelGGL(frn_pass_kernel<3>, grid, blk, 0, st, bt, W, logn, logL, (int)inverse, pr, po, post_n);
        ZK_HIP(hipGetLastError());
        logL += lr;
        in_b = !in_b;
    }
    return in_b;
}
static u32* frn_transform(const FrDomain& D, u32* a, u32* b, bool inverse, const u32* pre, const u32* post, u64 post_n, hipStream_t st) {
    return frn_transform_batch(D, &a, &b, 1, inverse, pre, post, post_n, st) ? b : a;
}

// EvaluationDomain::{fft, ifft, coset_fft, icoset_fft} on n = 2^logn elements of bellman's Fr (4 x u64 Montgomery,
// element-major), in place on the device
void FRN_FN(ntt_dev)(u64* d_data, int logn, bool inverse, bool coset, hipStream_t st) {
    const FrDomain& D = frn_domain(logn, st);
    const u64 n = 1ull << logn;
    DevBuf A, B;
    A.reserve(n * NR * 4); B.reserve(n * NR * 4);
    hipLaunchKernelGGL(frn_from_std_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const u32*)d_data, (u32*)A.p, n, n);
    const u32* pre = (!inverse && coset) ? (const u32*)D.GP.p : nullptr;
    const u32* post = inverse ? (coset ? (const u32*)D.GIP.p : (const u32*)D.minv()) : nullptr;
    u32* res = frn_transform(D, (u32*)A.p, (u32*)B.p, inverse, pre, post, inverse && !coset ? 1 : n, st);
    hipLaunchKernelGGL(frn_to_std_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const u32*)res, (u32*)d_data, n, n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(st));   // A, B go back to the pool
}

// prover.rs create_proof's `h` block on limb-major a, b, c (row evaluations, values < 2q): on return the buffer
// returned holds the n coefficients of (A B - C)/Z (the last one is zero for a satisfied system and is dropped by
// the caller).  t0..t2: scratch of the same size.
static u32* frn_quotient(const FrDomain& D, u32* a, u32* b, u32* c, u32* t0, u32* t1, u32* t2, hipStream_t st) {
    const u64 n = 1ull << D.logn;
    const u32 *GP = (const u32*)D.GP.p, *GIP = (const u32*)D.GIP.p, *minv = (const u32*)D.minv();
    u32* v[3] = {a, b, c};
    u32* s[3] = {t0, t1, t2};
    if (frn_transform_batch(D, v, s, 3, true, nullptr, minv, 1, st)) for (int i = 0; i < 3; ++i) std::swap(v[i], s[i]);   // ifft x 3
    if (frn_transform_batch(D, v, s, 3, false, GP, nullptr, n, st)) for (int i = 0; i < 3; ++i) std::swap(v[i], s[i]);    // coset_fft x 3
    hipLaunchKernelGGL(frn_quotient_pointwise_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, v[0], (const u32*)v[1], (const u32*)v[2], D.zinv(), n);
    ZK_HIP(hipGetLastError());
    return frn_transform(D, v[0], s[0], true, nullptr, GIP, n, st);                  // icoset_fft
}

// the same on bellman's element-major Montgomery vectors (a is overwritten with the n coefficients)
void FRN_FN(quotient_dev)(u64* d_a, const u64* d_b, const u64* d_c, int logn, hipStream_t st) {
    const FrDomain& D = frn_domain(logn, st);
    const u64 n = 1ull << logn;
    DevBuf buf[6];
    for (auto& x : buf) x.reserve(n * NR * 4);
    const u64* src[3] = {d_a, d_b, d_c};
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(frn_from_std_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const u32*)src[i], (u32*)buf[i].p, n, n);
    u32* res = frn_quotient(D, (u32*)buf[0].p, (u32*)buf[1].p, (u32*)buf[2].p, (u32*)buf[3].p, (u32*)buf[4].p, (u32*)buf[5].p, st);
    hipLaunchKernelGGL(frn_to_std_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, (const u32*)res, (u32*)d_a, n, n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(st));
}
