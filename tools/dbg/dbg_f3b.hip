#include "../../eigen-zkvm_amd/csrc/ntt_reg.hip.h"
#include <cstdio>
using gl::f3;
using namespace zk;
__device__ __forceinline__ f3 ld3(const u64* __restrict__ p) { return f3{{p[0], p[1], p[2]}}; }
__device__ __forceinline__ f3 mul_x(f3 a) { return f3{{a.v[2], gl::add(a.v[0], a.v[2]), a.v[1]}}; }
template <int LOGNX>
__global__ void kk(const u64* __restrict__ pol, u64 pol2_n, u64 shift_inv, u64 wi, u64 nx_inv, const u64* __restrict__ special_x, u64* __restrict__ out, u64* __restrict__ dbg) {
    constexpr int NX = 1 << LOGNX;
    const u64 g = threadIdx.x;
    if (g >= pol2_n) return;
    const u64 sinv = gl::mul(shift_inv, gl::pow(wi, g));
    const f3 y = gl::f3_muls(ld3(special_x), sinv);
    dbg[g * 16 + 0] = sinv; dbg[g * 16 + 1] = y.v[0]; dbg[g * 16 + 2] = y.v[1]; dbg[g * 16 + 3] = y.v[2];
    f3 S[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        u64 x[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = pol[((u64)i * pol2_n + g) * 3 + l];
        ntt_reg<LOGNX>(x, nullptr);
        f3 acc{{x[bitrev_c(NX - 1, LOGNX)], 0, 0}};
#pragma unroll
        for (int k = NX - 2; k >= 0; --k) {
            acc = gl::f3_mul(acc, y);
            acc.v[0] = gl::add(acc.v[0], x[bitrev_c(k, LOGNX)]);
        }
        S[l] = acc;
        dbg[g * 16 + 4 + 3 * l] = acc.v[0]; dbg[g * 16 + 5 + 3 * l] = acc.v[1]; dbg[g * 16 + 6 + 3 * l] = acc.v[2];
    }
    f3 r = gl::f3_add(S[0], gl::f3_add(mul_x(S[1]), mul_x(mul_x(S[2]))));
    r = gl::f3_muls(r, nx_inv);
    out[3 * g] = r.v[0]; out[3 * g + 1] = r.v[1]; out[3 * g + 2] = r.v[2];
}
static u64 hm(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL_P); }
int main() {
    const int n2 = 4;
    u64 pol[24] = {0}; for (int g = 0; g < n2; ++g) { pol[(0 * n2 + g) * 3] = 1; pol[(1 * n2 + g) * 3] = GL_P - 1; }
    u64 sx[3] = {10919585497513095254ULL, 12234714599883710599ULL, 13660377058527735992ULL};
    u64 si = gl::hpow(gl::hinv(49), 8), wi = gl::hinv(gl::hroot(3)), nxi = gl::hinv(2);
    u64 *dp, *ds, *dout, *ddbg; hipMalloc(&dp, 192); hipMalloc(&ds, 24); hipMalloc(&dout, 96); hipMalloc(&ddbg, 4 * 128);
    hipMemcpy(dp, pol, 192, hipMemcpyHostToDevice); hipMemcpy(ds, sx, 24, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kk<1>, dim3(1), dim3(256), 0, 0, dp, (u64)n2, si, wi, nxi, ds, dout, ddbg);
    u64 out[12], dbg[64]; hipMemcpy(out, dout, 96, hipMemcpyDeviceToHost); hipMemcpy(dbg, ddbg, 512, hipMemcpyDeviceToHost);
    for (int g = 0; g < n2; ++g) {
        u64 sinv = hm(si, gl::hpow(wi, g));
        printf("g=%d sinv ok=%d y ok=%d%d%d  S0=(%llx %llx %llx) exp 2y=(%llx %llx %llx) S1=(%llx %llx %llx) out=(%llx %llx %llx) exp y=(%llx %llx %llx)\n", g,
               dbg[g * 16] == sinv, dbg[g * 16 + 1] == hm(sx[0], sinv), dbg[g * 16 + 2] == hm(sx[1], sinv), dbg[g * 16 + 3] == hm(sx[2], sinv),
               dbg[g * 16 + 4], dbg[g * 16 + 5], dbg[g * 16 + 6], hm(2, hm(sx[0], sinv)), hm(2, hm(sx[1], sinv)), hm(2, hm(sx[2], sinv)),
               dbg[g * 16 + 7], dbg[g * 16 + 8], dbg[g * 16 + 9], out[3 * g], out[3 * g + 1], out[3 * g + 2], hm(sx[0], sinv), hm(sx[1], sinv), hm(sx[2], sinv));
    }
    return 0;
}
