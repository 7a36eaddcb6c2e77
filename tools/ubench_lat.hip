// Micro-benchmark: DEPENDENT-chain latency of the instructions a cooperative Poseidon permutation is made of, ONE wave on one SIMD
// (what a small proof's hash chains run as).  Prints cycles per instruction from s_memtime (shader clock).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lat.hip -o tools/ubench_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define N_REP 64

template <int WHICH>
__global__ __launch_bounds__(64) void k(u64* out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, b0 = a0 * 3 + 1, z = 0, idx = ((threadIdx.x + 1) & 15) * 4;
    u64 x0 = a0, c0 = b0;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b0));) }
        else if (WHICH == 1) { REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0" : "+v"(x0) : "v"(b0) : "s10", "s11");) }
        else if (WHICH == 2) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x0) : "v"(c0));) }
        else if (WHICH == 3) { REP64(asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n s_nop 1\n v_cndmask_b32_e64 %2, 0, 1, vcc\n v_add_u32 %0, %0, %2" : "+v"(a0), "+v"(b0), "+v"(z) : : "vcc");) }
        else if (WHICH == 4) { REP64(asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n s_nop 1\n v_subbrev_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a0) : "v"(b0) : "vcc");) }
        else if (WHICH == 5) { REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(idx));) }
        else if (WHICH == 6) { REP64(asm volatile("v_mov_b32 %0, %0" : "+v"(a0));) }
        else if (WHICH == 7) { REP64(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0));) }
        else if (WHICH == 8) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, -1, %0\n s_nop 1\n v_cndmask_b32_e64 %2, 0, -1, vcc\n v_add_u32 %1, %1, %2" : "+v"(x0), "+v"(b0), "+v"(z) : : "vcc");) }
        else if (WHICH == 9) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b0));) }
        else if (WHICH == 10) { REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %2, %0\n v_mad_u64_u32 %1, s[10:11], %2, %2, %1" : "+v"(x0), "+v"(c0) : "v"(b0) : "s10", "s11");) }   // two independent chains
        else if (WHICH == 11) { REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a0) : "v"(b0) : "vcc");) }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a0 + x0 + z + c0; }
}

template <int W> void run(const char* name, int per_rep) {
    u64* d; hipMalloc((void**)&d, 16);
    const int iters = 200;
    hipLaunchKernelGGL(k<W>, dim3(1), dim3(64), 0, 0, d, iters, 1u);
    hipLaunchKernelGGL(k<W>, dim3(1), dim3(64), 0, 0, d, iters, 2u);
    hipDeviceSynchronize();
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s %8.2f memtime-ticks per instruction (x%d per step)\n", name, (double)h[0] / ((double)iters * N_REP * per_rep), per_rep);
    hipFree(d);
}
int main() {
    // s_memtime ticks at a constant 100 MHz on this family; calibrate against a known chain: report ratios too
    run<6>("v_mov_b32 dependent", 1);
    run<0>("v_add_u32 dependent", 1);
    run<11>("v_add_co_u32 dependent (vcc out unused)", 1);
    run<9>("v_mul_lo_u32 dependent", 1);
    run<1>("v_mad_u64_u32 dependent (acc chain)", 1);
    run<10>("v_mad_u64_u32 two independent chains", 2);
    run<2>("v_lshl_add_u64 dependent", 1);
    run<3>("sub_co; nop1; cndmask; add (4 instr step)", 1);
    run<4>("sub_co; nop1; subbrev (3 instr step)", 1);
    run<8>("mad vcc; nop1; cndmask; add (4 instr step)", 1);
    run<7>("v_mov_b32_dpp dependent", 1);
    run<5>("ds_bpermute + waitcnt round trip", 1);
    return 0;
}
