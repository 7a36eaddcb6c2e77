// Radix-16 register transform, 64-bit modular butterflies (ntt_reg) against 24-bit limb butterflies (ntt_reg_limb):
// same values, time per transform.  hipcc -O3 --offload-arch=gfx950 -I eigen-zkvm_amd/csrc -I include tools/ubench/ubench_ntt16.hip
#include "ntt_limb.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int LOG, bool INV>
__global__ __launch_bounds__(256) void kern(u64* __restrict__ data, const u64* __restrict__ tw, int iters) {
    constexpr int n = 1 << LOG;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x[n], w[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { x[i] = data[t * n + i]; w[i] = tw[i]; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { ntt_reg<LOG, INV>(x);
#pragma unroll
            for (int i = 0; i < n; ++i) x[i] = gl::mul(x[i], w[i]); }
        if (MODE == 1) { ntt_reg_limb<LOG, INV, false>(x);
#pragma unroll
            for (int i = 0; i < n; ++i) x[i] = gl::mul(x[i], w[i]); }
        if (MODE == 2) ntt_reg<LOG, INV>(x);
        if (MODE == 3) ntt_reg_limb<LOG, INV, true>(x);
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < n; ++i) x[i] = gl::mul(x[i], w[i]); }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) data[t * n + i] = x[i];
}

template <int MODE, int LOG, bool INV>
double run(const std::vector<u64>& in, std::vector<u64>& out, const u64* d_tw, int iters, int blocks) {
    const size_t n = in.size();
    u64* d; CK(hipMalloc(&d, n * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(d, in.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((kern<MODE, LOG, INV>), dim3(blocks), dim3(256), 0, 0, d, d_tw, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    out.resize(n); CK(hipMemcpy(out.data(), d, n * 8, hipMemcpyDeviceToHost)); CK(hipFree(d));
    return best;
}

template <int LOG, bool INV>
void bench(int iters, int blocks) {
    constexpr int n = 1 << LOG;
    const size_t total = (size_t)blocks * 256 * n;
    std::vector<u64> in(total), tw(n), o0, o1, o2, o3, o4;
    u64 s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s % GL_P; };
    for (auto& v : in) v = rnd();
    for (size_t i = 0; i < 64 && i < total; ++i) in[i] = (i & 1) ? GL_P - 1 - (i >> 1) : (i >> 1);   // edge values
    for (auto& v : tw) v = rnd();
    u64* d_tw; CK(hipMalloc(&d_tw, n * 8)); CK(hipMemcpy(d_tw, tw.data(), n * 8, hipMemcpyHostToDevice));
    const double t0 = run<0, LOG, INV>(in, o0, d_tw, iters, blocks), t1 = run<1, LOG, INV>(in, o1, d_tw, iters, blocks);
    const double t2 = run<2, LOG, INV>(in, o2, d_tw, iters, blocks), t3 = run<3, LOG, INV>(in, o3, d_tw, iters, blocks);
    const double t4 = run<4, LOG, INV>(in, o4, d_tw, iters, blocks);
    size_t bad01 = 0, bad23 = 0;
    for (size_t i = 0; i < total; ++i) { bad01 += o0[i] != o1[i]; bad23 += o2[i] != o3[i]; }
    const double per = 1e6 / ((double)blocks * 256 * iters);   // ns per thread-transform... scaled below
    printf("radix-%d %s: 64-bit+mul %.3f ms | limb+mul %.3f ms | 64-bit %.3f ms | limb(canon) %.3f ms | mul only %.3f ms | mismatches %zu %zu\n",
           n, INV ? "inv" : "fwd", t0, t1, t2, t3, t4, bad01, bad23);
    (void)per;
    CK(hipFree(d_tw));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64, blocks = argc > 2 ? atoi(argv[2]) : 4096;
    bench<4, false>(iters, blocks); bench<4, true>(iters, blocks);
    bench<3, false>(iters, blocks); bench<3, true>(iters, blocks);
    bench<2, false>(iters, blocks); bench<2, true>(iters, blocks);
    return 0;
}
