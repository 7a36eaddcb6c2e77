// Poseidon-Goldilocks (t=12, x^7, 8 full + 22 partial rounds, "optimised" schedule), the
// LinearHash row digest and the binary Merkle tree built from them, for gfx950.
//
// Replaces starky/src/poseidon_opt.rs:80-200 (hash_inner), linearhash.rs:79-145 (hash/_hash),
// merklehash.rs:293-346 (merkelize), :79-134 (merklize_level), :47-61 (get_n_nodes).
//
// Mapping: one lane = one permutation (12-word state in 24 VGPRs).  The kernel is integer-ALU
// bound (2130 field multiplications per permutation, ~30 B of HBM traffic per permutation), so
// the work goes into the multiplier: round constants sit in __constant__ memory and are fetched
// with scalar loads (wave-uniform indices), and the dense MDS product exploits that M's entries
// are < 2^6: 12 64x6-bit partial products are accumulated in 128 bits and reduced ONCE per
// output word instead of 12 full modular multiplications.
#include "zk_internal.h"
#include "poseidon_gl_constants.h"

namespace zk {

namespace {

__constant__ u64 cC[118];
__constant__ u64 cM[144];
__constant__ u64 cP[144];
__constant__ u64 cS[506];

__device__ __forceinline__ u64 pow7(u64 x) {  // poseidon_opt.rs:68-74
    u64 x2 = gl::sqr(x), x3 = gl::mul(x2, x), x6 = gl::sqr(x3);
    return gl::mul(x6, x);
}

// out[i] = sum_j M[j][i] * st[j]  (poseidon_opt.rs:111-119), M[j][i] < 2^6
__device__ __forceinline__ void mds_small(u64 (&st)[12]) {
    u64 t[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        u64 lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const u32 m = (u32)cM[j * 12 + i];
            u64 pl = st[j] * m, ph = __umul64hi(st[j], (u64)m);
            lo += pl;
            hi += ph + (lo < pl);
        }
        t[i] = gl::reduce128(lo, hi);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = t[i];
}

__device__ __forceinline__ void mat_full(const u64* __restrict__ Mx, u64 (&st)[12]) {
    u64 t[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) acc = gl::add(acc, gl::mul(Mx[j * 12 + i], st[j]));
        t[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = t[i];
}

// in-place permutation of st = in[8] || cap[4]   (poseidon_opt.rs:98-199)
__device__ __noinline__ void poseidon_perm(u64 (&st)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = gl::add(st[i], cC[i]);
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = gl::add(pow7(st[i]), cC[(r + 1) * 12 + i]);
        mds_small(st);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = gl::add(pow7(st[i]), cC[48 + i]);
    mat_full(cP, st);
#pragma unroll 1
    for (int r = 0; r < 22; ++r) {
        st[0] = gl::add(pow7(st[0]), cC[60 + r]);
        u64 s0 = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) s0 = gl::add(s0, gl::mul(cS[23 * r + j], st[j]));
#pragma unroll
        for (int k = 1; k < 12; ++k) st[k] = gl::add(st[k], gl::mul(cS[23 * r + 11 + k], st[0]));
        st[0] = s0;
    }
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) st[i] = gl::add(pow7(st[i]), cC[82 + 12 * r + i]);
        mds_small(st);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = pow7(st[i]);
    mds_small(st);
}

// linearhash.rs:119-145 _hash over a global-memory segment: rate 8, capacity carried, tail
// zero-padded, <= 4 words -> identity padding.  Result in d[0..4).
__device__ __forceinline__ void sponge_gmem(const u64* __restrict__ v, u32 n, u64 (&d)[4]) {
    if (n <= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = (u32)i < n ? v[i] : 0;
        return;
    }
    u64 st[12];
#pragma unroll
    for (int i = 8; i < 12; ++i) st[i] = 0;
    for (u32 off = 0; off < n; off += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = (off + i < n) ? v[off + i] : 0;
        poseidon_perm(st);
#pragma unroll
        for (int i = 0; i < 4; ++i) st[8 + i] = st[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = st[i];
}

// linearhash.rs:79-110: bs = max(8, ceil(w/4)); <= 4 batch digests -> sponge over them
__device__ __forceinline__ void linearhash_row(const u64* __restrict__ row, u32 w, u64 (&out)[4]) {
    if (w <= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = (u32)i < w ? row[i] : 0;
        return;
    }
    u32 bs = (w + 3) / 4; if (bs < 8) bs = 8;
    const u32 hsz = (w + bs - 1) / bs;  // 1..4
    u64 h[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if ((u32)b < hsz) {
            const u32 len = (w - b * bs < bs) ? w - b * bs : bs;
            sponge_gmem(row + (u64)b * bs, len, h[b]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) h[b][i] = 0;
        }
    }
    if (hsz == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = h[0][i];
        return;
    }
    // sponge over hsz*4 (8, 12 or 16) digest words
    u64 st[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { st[i] = h[0][i]; st[4 + i] = h[1][i]; st[8 + i] = 0; }
    poseidon_perm(st);
    if (hsz > 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { st[8 + i] = st[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { st[i] = h[2][i]; st[4 + i] = h[3][i]; }  // h[3] is zero when hsz == 3
        poseidon_perm(st);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = st[i];
}

__global__ __launch_bounds__(256) void linearhash_rows_kernel(const u64* __restrict__ rows, u32 width, u64 height, u64* __restrict__ digests) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= height) return;
    u64 d[4];
    linearhash_row(rows + r * width, width, d);
#pragma unroll
    for (int i = 0; i < 4; ++i) digests[4 * r + i] = d[i];
}

// merklehash.rs:110-134 do_merklize_level: parent i = Poseidon(node[2i] || node[2i+1], cap 0)
__global__ __launch_bounds__(256) void merkle_level_kernel(const u64* __restrict__ in, u64 n_ops, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = in[8 * i + k];
#pragma unroll
    for (int k = 8; k < 12; ++k) st[k] = 0;
    poseidon_perm(st);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * i + k] = st[k];
}

// A tree over zero-width rows (tree2 / tree3 of a PIL without plookups or grand products,
// stark_gen.rs:311,359) has all-zero leaves, so every node of a level holds the same digest:
// one permutation per level instead of one per node.
__global__ void zero_tree_chain_kernel(u32 levels, u64* __restrict__ h /* [levels + 1][4] */) {
    if (threadIdx.x | blockIdx.x) return;
    u64 cur[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) h[k] = 0;
    for (u32 l = 0; l < levels; ++l) {
        u64 st[12];
        for (int k = 0; k < 4; ++k) { st[k] = cur[k]; st[4 + k] = cur[k]; st[8 + k] = 0; }
        poseidon_perm(st);
        for (int k = 0; k < 4; ++k) { cur[k] = st[k]; h[4 * (l + 1) + k] = st[k]; }
    }
}
__global__ void fill_digest_kernel(u64* __restrict__ nodes, u64 n, const u64* __restrict__ h) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * n) nodes[i] = h[i & 3];
}

__global__ void poseidon_one_kernel(const u64* in8, const u64* cap4, u64* out, int n_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 st[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = in8[k] >= GL_P ? in8[k] - GL_P : in8[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) st[8 + k] = cap4[k] >= GL_P ? cap4[k] - GL_P : cap4[k];
    poseidon_perm(st);
#pragma unroll
    for (int k = 0; k < 12; ++k) if (k < n_out) out[k] = st[k];
}

// ---- TranscriptGL (transcript.rs:8-103), device resident ---------------------------------------
// The Fiat-Shamir sponge lives in HBM next to the data it absorbs (roots, evals, final
// polynomial) and next to the kernels that consume its challenges, so a proof is one stream of
// launches with no host round trip until the query indices are needed.  One lane does the work
// (a sponge is inherently serial); state layout = TranscriptState below.
struct TranscriptState { u64 state[4]; u64 pending[8]; u64 out[12]; u32 n_pending, out_pos, n_out, _pad; };

__device__ void tr_update(TranscriptState* t) {  // transcript.rs:15-24
    u64 st[12];
    for (int i = 0; i < 8; ++i) st[i] = (u32)i < t->n_pending ? t->pending[i] : 0;
    for (int i = 0; i < 4; ++i) st[8 + i] = t->state[i];
    poseidon_perm(st);
    for (int i = 0; i < 12; ++i) t->out[i] = st[i];
    for (int i = 0; i < 4; ++i) t->state[i] = st[i];
    t->n_pending = 0; t->out_pos = 0; t->n_out = 12;
}
__device__ u64 tr_get1(TranscriptState* t) {     // transcript.rs:54-62
    if (t->out_pos >= t->n_out) tr_update(t);
    return t->out[t->out_pos++];
}
__global__ void tr_init_kernel(TranscriptState* t) {
    if (threadIdx.x | blockIdx.x) return;
    for (int i = 0; i < 4; ++i) t->state[i] = 0;
    t->n_pending = 0; t->out_pos = 0; t->n_out = 0;
}
__global__ void tr_put_kernel(TranscriptState* t, const u64* __restrict__ src, u64 n) {  // transcript.rs:25-33,64-71
    if (threadIdx.x | blockIdx.x) return;
    for (u64 i = 0; i < n; ++i) {
        t->n_out = 0; t->out_pos = 0;
        t->pending[t->n_pending++] = src[i];
        if (t->n_pending == 8) tr_update(t);
    }
}
__global__ void tr_get_kernel(TranscriptState* t, u64* __restrict__ dst, u32 n_words) {  // get_field = 3 words
    if (threadIdx.x | blockIdx.x) return;
    for (u32 i = 0; i < n_words; ++i) dst[i] = tr_get1(t);
}
__global__ void tr_permutations_kernel(TranscriptState* t, u32 n, u32 nbits, u64* __restrict__ dst) {  // transcript.rs:73-102
    if (threadIdx.x | blockIdx.x) return;
    u64 field = 0; u32 cur_bit = 63;  // force a fetch on first use
    for (u32 i = 0; i < n; ++i) {
        u64 a = 0;
        for (u32 j = 0; j < nbits; ++j) {
            if (cur_bit == 63) { field = tr_get1(t); cur_bit = 0; }
            if ((field >> cur_bit) & 1) a += 1ull << j;
            ++cur_bit;
        }
        dst[i] = a;
    }
}

bool g_consts_loaded[64] = {};

void ensure_constants() {
    int dev; ZK_HIP(hipGetDevice(&dev));
    ZK_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    if (g_consts_loaded[dev]) return;
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cC), ZK_POSEIDON_C, sizeof(ZK_POSEIDON_C)));
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cM), ZK_POSEIDON_M, sizeof(ZK_POSEIDON_M)));
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cP), ZK_POSEIDON_P, sizeof(ZK_POSEIDON_P)));
    ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cS), ZK_POSEIDON_S, sizeof(ZK_POSEIDON_S)));
    g_consts_loaded[dev] = true;
}

}  // namespace

size_t transcript_state_bytes() { return sizeof(TranscriptState); }
void transcript_init_dev(void* d_t, hipStream_t st) {
    ensure_constants();
    hipLaunchKernelGGL(tr_init_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t);
    ZK_HIP(hipGetLastError());
}
void transcript_put_dev(void* d_t, const u64* d_src, uint64_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(tr_put_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, d_src, n);
    ZK_HIP(hipGetLastError());
}
void transcript_get_dev(void* d_t, u64* d_dst, uint32_t n_words, hipStream_t st) {
    hipLaunchKernelGGL(tr_get_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, d_dst, n_words);
    ZK_HIP(hipGetLastError());
}
void transcript_permutations_dev(void* d_t, uint32_t n, uint32_t nbits, u64* d_dst, hipStream_t st) {
    ZK_REQUIRE(nbits >= 1 && nbits <= 63, "get_permutations: nbits out of range");
    hipLaunchKernelGGL(tr_permutations_kernel, dim3(1), dim3(64), 0, st, (TranscriptState*)d_t, n, nbits, d_dst);
    ZK_HIP(hipGetLastError());
}

void poseidon_dev(const u64* d_in8, const u64* d_cap4, u64* d_out, int n_out, hipStream_t st) {
    ensure_constants();
    hipLaunchKernelGGL(poseidon_one_kernel, dim3(1), dim3(64), 0, st, d_in8, d_cap4, d_out, n_out);
    ZK_HIP(hipGetLastError());
}

void linearhash_rows_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st) {
    ensure_constants();
    if (height == 0) return;
    const u64 blocks = (height + 255) / 256;
    hipLaunchKernelGGL(linearhash_rows_kernel, dim3((u32)blocks), dim3(256), 0, st, d_rows, width, height, d_digests);
    ZK_HIP(hipGetLastError());
}

uint64_t merkle_n_nodes(uint64_t n_) {  // merklehash.rs:47-61
    uint64_t n = n_, next_n = (n - 1) / 2 + 1, acc = next_n * 2;
    while (n > 1) {
        n = next_n; next_n = (n - 1) / 2 + 1;
        if (n > 1) acc += next_n * 2; else acc += 1;
    }
    return acc;
}

void merkelize_dev(const u64* d_rows, uint32_t width, uint64_t height, u64* d_nodes, hipStream_t st) {
    ensure_constants();
    ZK_REQUIRE(height >= 1, "merkelize: height must be >= 1");
    const uint64_t nn = merkle_n_nodes(height);
    // absent right siblings on odd levels are the all-zero digest (merklehash.rs:307)
    ZK_HIP(hipMemsetAsync(d_nodes, 0, nn * 32, st));
    if (width == 0 && (height & (height - 1)) == 0 && height > 1) {  // all-zero leaves, full binary tree
        uint32_t levels = 0;
        while ((1ull << levels) < height) ++levels;
        DevBuf hbuf;  // pooled; returned at scope exit, reuse is stream ordered
        hbuf.reserve((levels + 1) * 32);
        u64* d_h = hbuf.u();
        hipLaunchKernelGGL(zero_tree_chain_kernel, dim3(1), dim3(64), 0, st, levels, d_h);
        ZK_HIP(hipGetLastError());
        uint64_t n = height, off = 0;
        for (uint32_t l = 1; l <= levels; ++l) {  // level l has height >> l nodes, starting after level l-1
            off += n; n >>= 1;
            hipLaunchKernelGGL(fill_digest_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, st, d_nodes + 4 * off, n, d_h + 4 * l);
            ZK_HIP(hipGetLastError());
        }
        return;
    }
    linearhash_rows_dev(d_rows, width, height, d_nodes, st);
    uint64_t n64 = height, next = (n64 - 1) / 2 + 1, p_in = 0, p_out = next * 2;
    while (n64 > 1) {  // merklehash.rs:331-343
        const u64 blocks = (next + 255) / 256;
        hipLaunchKernelGGL(merkle_level_kernel, dim3((u32)blocks), dim3(256), 0, st, d_nodes + 4 * p_in, next, d_nodes + 4 * p_out);
        ZK_HIP(hipGetLastError());
        n64 = next; next = (n64 - 1) / 2 + 1; p_in = p_out; p_out = p_in + next * 2;
    }
}

}  // namespace zk
