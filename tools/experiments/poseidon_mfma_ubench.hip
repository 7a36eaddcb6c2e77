// Stage A of the round-6 experiment: Poseidon-GL's dense 12 x 12 product (the pre-sparse matrix P) on the vector pipe (mat_full, as shipped
// through round 5) against the matrix-pipe form (csrc/gl_mfma.hip.h), one permutation per lane, three waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I eigen-zkvm_amd/csrc tools/experiments/poseidon_mfma_ubench.hip -o tools/experiments/poseidon_mfma_ubench \
//         -L eigen-zkvm_amd -l:libzkgpu.so -Wl,-rpath,'$ORIGIN/../../eigen-zkvm_amd'
// Prints: layout probe, equality of both forms with the host's 128-bit arithmetic, time per product.  SQ_INSTS_VALU: run under
// rocprofv3 --pmc SQ_INSTS_VALU (tools/gpu_round.sh mfma_pmc).
#define ZK_POSEIDON_MFMA_UBENCH 1
#include "poseidon.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace zk;

__device__ u64 g_mfma_tab[pmfma::TAB_WORDS];

template <int ITERS>
__global__ __launch_bounds__(256, 3) void k_valu(const u64* __restrict__ in, u64* __restrict__ out) {
    ZK_POSEIDON_LDS;
    load_tables(tab);
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = in[12 * r + i];
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) mat_full(tab + T_PT, st);
#pragma unroll
    for (int i = 0; i < 12; ++i) out[12 * r + i] = st[i] >= GL_P ? st[i] - GL_P : st[i];
}
template <int ITERS>
__global__ __launch_bounds__(256, 3) void k_mfma(const u64* __restrict__ in, u64* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) u64 mt[pmfma::TAB_WORDS];
    for (int i = threadIdx.x; i < pmfma::TAB_WORDS; i += blockDim.x) mt[i] = g_mfma_tab[i];
    __syncthreads();
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = in[12 * r + i];
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        pmfma::BOps B;
        pmfma::make_b<12>(B, [&](int j) { return st[j]; });
        pmfma::product<3>(B, mt, [&](int o, u64 v) { st[o] = v; });
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) out[12 * r + i] = st[i] >= GL_P ? st[i] - GL_P : st[i];
}
__global__ void k_probe(u32* o) {
    u32 x = threadIdx.x, y = 1000 + threadIdx.x;
    pmfma::swap32(x, y);
    o[threadIdx.x] = x; o[64 + threadIdx.x] = y;
}

static u64 hmulmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL_P); }

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 3 * 8;
    const size_t n = (size_t)blocks * 256;
    ensure_constants();
    static u64 mtab[pmfma::TAB_WORDS];
    u64 coef[144];
    for (int o = 0; o < 12; ++o) for (int j = 0; j < 12; ++j) coef[o * 12 + j] = ZK_POSEIDON_P[12 * j + o];   // out[o] = sum_j P[j][o] st[j]
    if (!pmfma::build_tables(coef, 12, 12, nullptr, mtab)) { printf("build_tables: out of range\n"); return 1; }
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_tab), mtab, sizeof(mtab)));

    u32* d_p; ZK_HIP(hipMalloc((void**)&d_p, 128 * 4));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_p);
    u32 hp[128]; ZK_HIP(hipMemcpy(hp, d_p, sizeof(hp), hipMemcpyDeviceToHost));
    printf("permlane32_swap(x = lane, y = 1000 + lane): x' lanes 0,31,32,63 = %u %u %u %u; y' = %u %u %u %u\n", hp[0], hp[31], hp[32], hp[63], hp[64], hp[95], hp[96], hp[127]);

    std::vector<u64> h_in(12 * n), h_a(12 * n), h_b(12 * n);
    u64 s = 0x9E3779B97F4A7C15ull;
    for (auto& v : h_in) { s += 0x9E3779B97F4A7C15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; v = z ^ (z >> 31); }
    // a few extreme words
    for (int i = 0; i < 12; ++i) { h_in[i] = ~0ull; h_in[12 + i] = 0; h_in[24 + i] = GL_P - 1; h_in[36 + i] = 0x8080808080808080ull; h_in[48 + i] = 0x7F7F7F7F7F7F7F7Full; }
    u64 *d_in, *d_out;
    ZK_HIP(hipMalloc((void**)&d_in, 96 * n)); ZK_HIP(hipMalloc((void**)&d_out, 96 * n));
    ZK_HIP(hipMemcpy(d_in, h_in.data(), 96 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out);
    ZK_HIP(hipMemcpy(h_a.data(), d_out, 96 * n, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out);
    ZK_HIP(hipMemcpy(h_b.data(), d_out, 96 * n, hipMemcpyDeviceToHost));
    size_t bad_a = 0, bad_b = 0, shown = 0;
    for (size_t r = 0; r < n && r < 65536; ++r)
        for (int o = 0; o < 12; ++o) {
            u64 want = 0;
            for (int j = 0; j < 12; ++j) { const u64 t = hmulmod(coef[o * 12 + j], h_in[12 * r + j] % GL_P); want = (u64)(((unsigned __int128)want + t) % GL_P); }
            if (h_a[12 * r + o] != want) ++bad_a;
            if (h_b[12 * r + o] != want) { ++bad_b; if (shown++ < 8) printf("  mfma mismatch row %zu out %d: got %016llx want %016llx\n", r, o, (unsigned long long)h_b[12 * r + o], (unsigned long long)want); }
        }
    size_t diff = 0;
    for (size_t i = 0; i < 12 * n; ++i) diff += h_a[i] != h_b[i];
    printf("vs host (first 65536 rows): valu mismatches %zu, mfma mismatches %zu; valu vs mfma over all %zu rows: %zu words differ\n", bad_a, bad_b, n, diff);

    hipEvent_t e0, e1; ZK_HIP(hipEventCreate(&e0)); ZK_HIP(hipEventCreate(&e1));
    constexpr int IT = 64;
    for (int rep = 0; rep < 3; ++rep) {
        float ma, mb;
        ZK_HIP(hipEventRecord(e0)); hipLaunchKernelGGL(k_valu<IT>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); ZK_HIP(hipEventRecord(e1)); ZK_HIP(hipEventSynchronize(e1));
        ZK_HIP(hipEventElapsedTime(&ma, e0, e1));
        ZK_HIP(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma<IT>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); ZK_HIP(hipEventRecord(e1)); ZK_HIP(hipEventSynchronize(e1));
        ZK_HIP(hipEventElapsedTime(&mb, e0, e1));
        printf("rep %d: %zu lanes x %d products: valu %.3f ms (%.2f ns/product/lane-batch), mfma %.3f ms  ratio %.3f\n", rep, n, IT, ma, ma * 1e6 / IT / n, mb, mb / ma);
    }
    ZK_HIP(hipMemcpy(h_a.data(), d_out, 96 * n, hipMemcpyDeviceToHost));
    return bad_b || diff ? 2 : 0;
}
