// Probe: throughput of candidate Goldilocks field-op implementations on gfx950 in an
// NTT-shaped register workload (16 values/thread, radix-16 DIF + twiddle multiply, repeated).
// All policies must produce identical canonical outputs (checked against policy 0).
// Build: hipcc --offload-arch=gfx950 -O3 tools/field_probe.hip -o tools/field_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define P 0xFFFFFFFF00000001ULL
#define EPS 0xFFFFFFFFULL

// ---------------- policy 0: current product code (64-bit compare style) ----------------------
struct F0 {
    static __device__ __forceinline__ u64 add(u64 a, u64 b) { u64 s = a + b; return (s < a || s >= P) ? s - P : s; }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b) { u64 d = a - b; return a < b ? d + P : d; }
    static __device__ __forceinline__ u64 red(u64 lo, u64 hi) {
        u64 hh = hi >> 32, hl = hi & EPS;
        u64 t0 = lo - hh; if (lo < hh) t0 -= EPS;
        u64 t1 = (hl << 32) - hl;
        u64 t2 = t0 + t1; if (t2 < t1) t2 += EPS;
        return t2 >= P ? t2 - P : t2;
    }
    static __device__ __forceinline__ u64 mul(u64 a, u64 b) { return red(a * b, __umul64hi(a, b)); }
};

// ---------------- policy 1: explicit 4-mad product, same reduction ---------------------------
struct F1 : F0 {
    static __device__ __forceinline__ u64 mul(u64 a, u64 b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 p0 = (u64)a0 * b0;
        u64 p1 = (u64)a0 * b1 + (p0 >> 32);
        u64 p2 = (u64)a1 * b0 + (u32)p1;
        u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
        return red((p2 << 32) | (u32)p0, p3);
    }
};

// ---------------- policy 2: 32-bit limb carry chains (clang addc/subc builtins) ---------------
struct F2 {
    static __device__ __forceinline__ u64 mk(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
    static __device__ __forceinline__ u64 add(u64 a, u64 b) {
        u32 c0, c1, d0, d1;
        u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
        u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
        u32 t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &d0);
        u32 t1 = __builtin_addc(s1, 0u, d0, &d1);
        bool sel = (c1 | d1) != 0;
        return mk(sel ? t0 : s0, sel ? t1 : s1);
    }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b) {
        u32 b0, b1, e0, e1;
        u32 d0 = __builtin_subc((u32)a, (u32)b, 0u, &b0);
        u32 d1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b0, &b1);
        u32 m = 0u - b1;
        u32 r0 = __builtin_subc(d0, m, 0u, &e0);
        u32 r1 = __builtin_subc(d1, 0u, e0, &e1);
        return mk(r0, r1);
    }
    static __device__ __forceinline__ u64 mul(u64 a, u64 b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 p0 = (u64)a0 * b0;
        u64 p1 = (u64)a0 * b1 + (p0 >> 32);
        u64 p2 = (u64)a1 * b0 + (u32)p1;
        u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
        u32 r0 = (u32)p0, r1 = (u32)p2, r2 = (u32)p3, r3 = (u32)(p3 >> 32);
        // S1: t = (r1:r0) - r3, fix borrow by -EPS
        u32 bw0, bw1, e0, e1;
        u32 t0 = __builtin_subc(r0, r3, 0u, &bw0);
        u32 t1 = __builtin_subc(r1, 0u, bw0, &bw1);
        u32 m = 0u - bw1;
        t0 = __builtin_subc(t0, m, 0u, &e0);
        t1 = __builtin_subc(t1, 0u, e0, &e1);
        // S2: u = t + r2*EPS (mad), fix carry by +EPS
        u64 t = mk(t0, t1);
        u64 u = t + (u64)r2 * EPS;
        u32 cm = (u < t) ? 0xFFFFFFFFu : 0u;
        u32 c0, c1;
        u32 u0 = __builtin_addc((u32)u, cm, 0u, &c0);
        u32 u1 = __builtin_addc((u32)(u >> 32), 0u, c0, &c1);
        // S3: canonicalize
        u32 d0, d1;
        u32 v0 = __builtin_addc(u0, 0xFFFFFFFFu, 0u, &d0);
        u32 v1 = __builtin_addc(u1, 0u, d0, &d1);
        return mk(d1 ? v0 : u0, d1 ? v1 : u1);
    }
};

// ---------------- policy 3: inline-asm multiplier (mad carry-out), F2 add/sub ----------------
struct F3 : F2 {
    static __device__ __forceinline__ u64 mul(u64 a, u64 b) {
        u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
        u64 p0 = (u64)a0 * b0;
        u64 p1 = (u64)a0 * b1 + (p0 >> 32);
        u64 p2 = (u64)a1 * b0 + (u32)p1;
        u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
        u64 lo = (p2 << 32) | (u32)p0;
        u32 r2 = (u32)p3, r3 = (u32)(p3 >> 32);
        u32 bw0, bw1, e0, e1;
        u32 t0 = __builtin_subc((u32)lo, r3, 0u, &bw0);
        u32 t1 = __builtin_subc((u32)(lo >> 32), 0u, bw0, &bw1);
        u32 mb = 0u - bw1;
        t0 = __builtin_subc(t0, mb, 0u, &e0);
        t1 = __builtin_subc(t1, 0u, e0, &e1);
        u64 t = mk(t0, t1), res; u32 m;
        asm volatile("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, vcc"
                     : "=&v"(res), "=v"(m) : "v"(r2), "v"(t) : "vcc");
        res += m;
        // canonicalize
        u32 d0, d1;
        u32 v0 = __builtin_addc((u32)res, 0xFFFFFFFFu, 0u, &d0);
        u32 v1 = __builtin_addc((u32)(res >> 32), 0u, d0, &d1);
        return mk(d1 ? v0 : (u32)res, d1 ? v1 : (u32)(res >> 32));
    }
};

__host__ __device__ constexpr int brev4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

template <class F>
__device__ __forceinline__ void ntt16(u64 (&x)[16], const u64* __restrict__ w) {
#pragma unroll
    for (int lh = 3; lh >= 0; --lh) {
        const int half = 1 << lh;
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half)
#pragma unroll
            for (int j = 0; j < half; ++j) {
                u64 a = x[blk + j], b = x[blk + j + half];
                x[blk + j] = F::add(a, b);
                u64 d = F::sub(a, b);
                x[blk + j + half] = (j == 0) ? d : F::mul(d, w[j * (8 / half)]);
            }
    }
}

template <class F>
__global__ __launch_bounds__(256) void probe(const u64* __restrict__ in, u64* __restrict__ out, const u64* __restrict__ w, int iters) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = in[gid * 16 + i];
    for (int it = 0; it < iters; ++it) {
        ntt16<F>(x, w);
#pragma unroll
        for (int i = 1; i < 16; ++i) x[i] = F::mul(x[i], w[8 + i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[gid * 16 + i] = x[i];
}

static u64 hmul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % P); }
static u64 hpow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = hmul(r, a); a = hmul(a, a); e >>= 1; } return r; }

template <class F>
double run(const char* name, const u64* d_in, u64* d_out, const u64* d_w, size_t nthreads, int iters, std::vector<u64>* keep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<F>, dim3(nthreads / 256), dim3(256), 0, 0, d_in, d_out, d_w, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<F>, dim3(nthreads / 256), dim3(256), 0, 0, d_in, d_out, d_w, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<u64> h(nthreads * 16);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    bool same = true;
    if (keep->empty()) *keep = h; else same = (h == *keep);
    double per_thread_iter_cycles = ms * 1e-3 * 2.4e9 / ((double)nthreads / 64 / 1024 * iters);  // cycles per wave-iteration per SIMD
    printf("%-34s %8.3f ms  %7.0f cycles/(16-pt DFT + 15 tw muls) per wave  match=%d\n", name, ms, per_thread_iter_cycles, (int)same);
    return ms;
}

int main() {
    const size_t nthreads = 256 * 4 * 64 * 5;  // 5 waves per SIMD
    const int iters = 200;
    std::vector<u64> h_in(nthreads * 16), h_w(24);
    u64 s = 88172645463325252ULL;
    for (auto& v : h_in) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s % P; }
    h_in[0] = 0; h_in[1] = P - 1; h_in[2] = 1; h_in[3] = EPS; h_in[4] = P - EPS; h_in[5] = 1ULL << 32;
    u64 w16 = hpow(hpow(7, 0xFFFFFFFFULL), 1ULL << 28);  // MG[4]
    u64 c = 1;
    for (int i = 0; i < 8; ++i) { h_w[i] = c; c = hmul(c, w16); }
    for (int i = 8; i < 24; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h_w[i] = s % P; }
    u64 *d_in, *d_out, *d_w;
    (void)hipMalloc(&d_in, h_in.size() * 8); (void)hipMalloc(&d_out, h_in.size() * 8); (void)hipMalloc(&d_w, h_w.size() * 8);
    (void)hipMemcpy(d_in, h_in.data(), h_in.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_w, h_w.data(), h_w.size() * 8, hipMemcpyHostToDevice);
    std::vector<u64> keep;
    run<F0>("F0 current", d_in, d_out, d_w, nthreads, iters, &keep);
    run<F1>("F1 explicit 4-mad product", d_in, d_out, d_w, nthreads, iters, &keep);
    run<F2>("F2 32-bit limb builtins", d_in, d_out, d_w, nthreads, iters, &keep);
    run<F3>("F3 asm reduce + F2 add/sub", d_in, d_out, d_w, nthreads, iters, &keep);
    return 0;
}
