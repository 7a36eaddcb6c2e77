// Internal plumbing shared by the HIP translation units of libzkgpu (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <stdexcept>
#include "gl.cuh"

namespace zk {

// ---- error handling: C ABI returns int status, message kept per thread (include/zkgpu.h) ----
void set_error(const std::string& msg);
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define ZK_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            throw zk::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +     \
                            __FILE__ + ":" + std::to_string(__LINE__) + ")");                     \
    } while (0)

#define ZK_REQUIRE(cond, msg)                                                                     \
    do { if (!(cond)) throw zk::Error(std::string(msg)); } while (0)

// ---- device buffers ----
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t n) {  // grow-only workspace
        if (n <= bytes) return;
        if (p) { ZK_HIP(hipFree(p)); p = nullptr; bytes = 0; }
        ZK_HIP(hipMalloc(&p, n)); bytes = n;
    }
    u64* u() const { return (u64*)p; }
};

// ---- NTT (ntt.hip) ----
// natural-order batched NTT over a row-major [1<<nbits][n_pols] device matrix; dst != src.
// `tmp` must hold (1<<nbits)*n_pols words when the plan has more than one pass.
void ntt_dev(const u64* d_src, u64* d_dst, u64* d_tmp, uint32_t n_pols, uint32_t nbits, bool inverse, hipStream_t st);
// low-degree extension on the coset 49*<w_ext>: [1<<nbits][n_pols] -> [1<<nbits_ext][n_pols].
// tmp: (1<<nbits_ext)*n_pols words.
void lde_dev(const u64* d_src, u64* d_dst, u64* d_tmp, uint32_t n_pols, uint32_t nbits, uint32_t nbits_ext, hipStream_t st);
int ntt_num_passes(uint32_t nbits);

// ---- Poseidon / Merkle (poseidon.hip) ----
void poseidon_dev(const u64* d_in8, const u64* d_cap4, u64* d_out, int n_out, hipStream_t st);
void linearhash_rows_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st);
uint64_t merkle_n_nodes(uint64_t height);
// nodes: merkle_n_nodes(height)*4 words, zero-filled by this call; leaves from [height][width] rows
void merkelize_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st);

}  // namespace zk
