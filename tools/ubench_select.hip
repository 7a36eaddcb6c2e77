// Micro-benchmark: what a conditional select costs on gfx950, by where its condition lives (VCC through the VOP2 encoding,
// an SGPR pair through VOP3) and by who wrote the condition (VALU compare, carry-out of an add, SALU move).
// Prints cycles per wave-instruction per SIMD at a nominal 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_select.hip -o tools/ubench_select
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
    if (WHICH == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(b0) : "vcc");
    if (WHICH == 1) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
    if (WHICH == 2) asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11");
    if (WHICH == 3) asm volatile("v_cmp_lt_u32 s[10:11], %0, %1" ::"v"(a0), "v"(b0) : "s10", "s11");
    if (WHICH == 4) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(b0) : "vcc");
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0 || WHICH == 1) {  // VOP2, condition in VCC, written once
            REP64(asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cndmask_b32_e32 %2, %2, %6, vcc\n v_cndmask_b32_e32 %3, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 2 || WHICH == 3) {  // VOP3, condition in s[10:11], written once
            REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %5, s[10:11]\n v_cndmask_b32_e64 %2, %2, %6, s[10:11]\n v_cndmask_b32_e64 %3, %3, %7, s[10:11]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 4) {  // VOP3 encoding, condition in VCC
            REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %5, vcc\n v_cndmask_b32_e64 %2, %2, %6, vcc\n v_cndmask_b32_e64 %3, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 5) {  // compare into VCC, two fillers, two selects (a 64-bit select as the compiler writes it)
            REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n v_cndmask_b32_e32 %0, %0, %5, vcc\n v_cndmask_b32_e32 %1, %1, %6, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 6) {  // the same through an SGPR pair
            REP64(asm volatile("v_cmp_lt_u32 s[10:11], %0, %4\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n v_cndmask_b32_e64 %0, %0, %5, s[10:11]\n v_cndmask_b32_e64 %1, %1, %6, s[10:11]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s10", "s11");)
        } else if (WHICH == 7) {  // five plain adds (reference for 5 and 6)
            REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %6"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (WHICH == 8) {  // modular add as gl::add compiles: add_co, addc_co, cndmask of the correction, add_co, addc
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n s_nop 1\n v_cndmask_b32_e64 %2, 0, -1, vcc\n"
                               "v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, 0, %1, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        } else if (WHICH == 9) {  // the same with the carry kept in an SGPR pair
            REP64(asm volatile("v_add_co_u32 %0, s[10:11], %0, %4\n v_addc_co_u32 %1, s[10:11], %1, %5, s[10:11]\n s_nop 1\n v_cndmask_b32_e64 %2, 0, -1, s[10:11]\n"
                               "v_add_co_u32 %0, s[12:13], %0, %2\n v_addc_co_u32 %1, s[12:13], 0, %1, s[12:13]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s10", "s11", "s12", "s13");)
        } else if (WHICH == 10) {  // v_subbrev / borrow as mask: sub_co, subb_co, then mask = 0 - borrow via v_subb (no cndmask)
            REP64(asm volatile("v_sub_co_u32 %0, vcc, %0, %4\n v_subb_co_u32 %1, vcc, %1, %5, vcc\n s_nop 1\n v_subb_co_u32 %2, vcc, %2, %2, vcc\n"
                               "v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, 0, %1, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int W>
void run(const char* name, u64* d, int waves_per_simd, int per_rep) {
    const int iters = 2000;
    dim3 grid(256 * waves_per_simd), block(256);  // 256-thread blocks: 1 wave per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, grid, block, 0, 0, d, iters, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, grid, block, 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)iters * 64 * per_rep * waves_per_simd;  // VALU wave-instructions per SIMD
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-58s waves/SIMD=%d  %7.3f ms  -> %5.2f cycles per VALU wave-instruction per SIMD\n", name, waves_per_simd, ms, cycles / winst);
}

int main() {
    u64* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(u64));
    for (int w : {8, 4, 1}) {
        run<0>("cndmask e32 vcc (vcc from v_cmp, once)", d, w, 4);
        run<1>("cndmask e32 vcc (vcc from s_mov, once)", d, w, 4);
        run<2>("cndmask e64 s[10:11] (from s_mov, once)", d, w, 4);
        run<3>("cndmask e64 s[10:11] (from v_cmp, once)", d, w, 4);
        run<4>("cndmask e64 vcc (from v_cmp, once)", d, w, 4);
        run<5>("v_cmp vcc; 2 adds; 2 cndmask e32 vcc", d, w, 5);
        run<6>("v_cmp sgpr; 2 adds; 2 cndmask e64 sgpr", d, w, 5);
        run<7>("5 x v_add_u32", d, w, 5);
        run<8>("mod add: add_co,addc,nop,cndmask vcc,add_co,addc", d, w, 5);
        run<9>("mod add with sgpr-pair carries", d, w, 5);
        run<10>("mod sub: sub_co,subb,nop,subb(mask),add_co,addc", d, w, 5);
    }
    return 0;
}
