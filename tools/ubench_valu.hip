// Micro-benchmark: issue cost of the integer VALU instructions a Goldilocks multiplier is built
// from, on gfx950.  Prints cycles per wave-instruction per SIMD (assuming 2.4 GHz, 1024 SIMDs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int WHICH>
__global__ __launch_bounds__(256) void k(u64* out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    u32 b0 = a0 ^ 0x55, b1 = a1 ^ 0x77, b2 = a2 ^ 0x99, b3 = a3 ^ 0xbb;
    u64 x0 = a0, x1 = a1, x2 = a2, x3 = a3;
    u64 c0 = b0, c1 = b1, c2 = b2, c3 = b3;
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) {  // v_add_u32 (baseline full-rate), 4 independent chains
            REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 1) {  // v_mad_u64_u32
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %5, %6, %1\n"
                               "v_mad_u64_u32 %2, s[10:11], %6, %7, %2\n v_mad_u64_u32 %3, s[10:11], %7, %4, %3"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s10", "s11");)
        } else if (WHICH == 2) {  // v_mul_lo_u32
            REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 3) {  // v_mul_hi_u32
            REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %6\n v_mul_hi_u32 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 4) {  // v_lshl_add_u64 (64-bit add)
            REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %6\n v_lshl_add_u64 %3, %3, 0, %7"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c0), "v"(c1), "v"(c2), "v"(c3));)
        } else if (WHICH == 5) {  // add_co / addc pairs through vcc, interleaved 2 chains (hazard visible?)
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n"
                               "v_add_co_u32 %2, vcc, %2, %6\n v_addc_co_u32 %3, vcc, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 6) {  // v_mul_u32_u24
            REP64(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 7) {  // v_cmp_lt_u64 + cndmask x2 through sgpr pair
            REP64(asm volatile("v_cmp_lt_u64 s[10:11], %0, %2\n v_cmp_lt_u64 s[12:13], %1, %3\n"
                               "v_cndmask_b32 %4, %4, %5, s[10:11]\n v_cndmask_b32 %6, %6, %7, s[12:13]"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s10", "s11", "s12", "s13");)
        } else if (WHICH == 8) {  // v_add3_u32
            REP64(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 9) {  // v_mad_u32_u24
            REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %5, %6\n v_mad_u32_u24 %2, %2, %6, %7\n v_mad_u32_u24 %3, %3, %7, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 10) {  // v_lshlrev_b64
            REP64(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 7, %2\n v_lshlrev_b64 %3, 9, %3"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if (WHICH == 12) {  // add_co / addc with the 2 wait states the compiler inserts
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n"
                               "v_add_co_u32 %2, vcc, %2, %6\n s_nop 1\n v_addc_co_u32 %3, vcc, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 13) {  // two carry chains interleaved through distinct sgpr pairs (1 instr gap)
            REP64(asm volatile("v_add_co_u32 %0, s[10:11], %0, %4\n v_add_co_u32 %2, s[12:13], %2, %6\n s_nop 0\n"
                               "v_addc_co_u32 %1, s[10:11], %1, %5, s[10:11]\n v_addc_co_u32 %3, s[12:13], %3, %7, s[12:13]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s10", "s11", "s12", "s13");)
        } else if (WHICH == 14) {
            REP64(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 15) {
            REP64(asm volatile("v_and_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_or_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 16) {
            REP64(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 5, %2\n v_lshrrev_b32 %3, 2, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (WHICH == 17) {
            REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 18) {  // cndmask e32 (vcc set once)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(b0) : "vcc");
            REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 19) {  // add_co e32 independent (write vcc only)
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %5\n v_add_co_u32 %2, vcc, %2, %6\n v_add_co_u32 %3, vcc, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 20) {  // v_add_u32 forced VOP3 encoding
            REP64(asm volatile("v_add_u32_e64 %0, %0, %4\n v_add_u32_e64 %1, %1, %5\n v_add_u32_e64 %2, %2, %6\n v_add_u32_e64 %3, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 21) {  // add_co, 2 filler adds, addc (hazard-free carry chain)
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n v_addc_co_u32 %1, vcc, %1, %5, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 22) {  // v_lshl_add_u32 / v_lshl_or_b32 (VOP3, 32-bit)
            REP64(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %5\n v_lshl_add_u32 %2, %2, 1, %6\n v_lshl_or_b32 %3, %3, 1, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 23) {  // v_mad_u64_u32 with inline-constant multiplier (zext-add form)
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %4, 1, %0\n v_mad_u64_u32 %1, s[10:11], %5, 1, %1\n"
                               "v_mad_u64_u32 %2, s[10:11], %6, -1, %2\n v_mad_u64_u32 %3, s[10:11], %7, -1, %3"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s10", "s11");)
        } else if (WHICH == 24) {  // v_mov_b64
            REP64(asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c0), "v"(c1), "v"(c2), "v"(c3));)
        } else if (WHICH == 25) {  // v_cmp_lt_u32 e32 (vcc) independent
            REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %5\n v_cmp_lt_u32 vcc, %2, %6\n v_cmp_lt_u32 vcc, %3, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 11) {  // v_alignbit_b32
            REP64(asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %5, 7\n v_alignbit_b32 %2, %2, %6, 7\n v_alignbit_b32 %3, %3, %7, 7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + x0 + x1 + x2 + x3;
}

template <int W>
void run(const char* name, u64* d, int waves_per_simd) {
    const int iters = 3000;
    dim3 grid(256 * waves_per_simd), block(256);  // 256-thread blocks: 1 wave per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, grid, block, 0, 0, d, 3000, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, grid, block, 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)iters * 64 * 4 * waves_per_simd;  // wave-instructions per SIMD
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cycles / winst);
}

int main() {
    u64* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(u64));
    for (int w : {8}) {
        run<0>("v_add_u32", d, w);
        run<1>("v_mad_u64_u32", d, w);
        run<2>("v_mul_lo_u32", d, w);
        run<3>("v_mul_hi_u32", d, w);
        run<4>("v_lshl_add_u64", d, w);
        run<5>("v_add_co+v_addc (vcc)", d, w);
        run<6>("v_mul_u32_u24", d, w);
        run<7>("v_cmp_lt_u64+cndmask", d, w);
        run<8>("v_add3_u32", d, w);
        run<9>("v_mad_u32_u24", d, w);
        run<10>("v_lshlrev_b64", d, w);
        run<11>("v_alignbit_b32", d, w);
        run<12>("add_co+s_nop1+addc", d, w);
        run<13>("2 chains sgpr carry", d, w);
        run<14>("v_sub_u32", d, w);
        run<15>("v_and/xor/or_b32", d, w);
        run<16>("v_lshl/lshr_b32", d, w);
        run<17>("v_mov_b32", d, w);
        run<18>("v_cndmask_b32 e32 vcc", d, w);
        run<19>("v_add_co_u32 e32 indep", d, w);
        run<20>("v_add_u32_e64", d, w);
        run<21>("add_co,2 fill,addc", d, w);
        run<22>("v_lshl_add_u32/lshl_or", d, w);
        run<23>("v_mad_u64_u32 x*1/-1+acc", d, w);
        run<24>("v_mov_b64", d, w);
        run<25>("v_cmp_lt_u32 e32", d, w);
    }
    return 0;
}
