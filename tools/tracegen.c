/* Synthetic input generator (not oracle, not product): "wide Fibonacci" execution trace for
 * tools/synth_pil.py -- W independent pairs (a_k, b_k) per row, a' = b, b' = a + b (mod p),
 * row-major [N][2W] little-endian u64, plus splitmix64 columns for the NTT benchmark.
 * Build: gcc -O2 -shared -fPIC tools/tracegen.c -o tools/libtracegen.so */
#include <stdint.h>
#define P 0xFFFFFFFF00000001ULL
static inline uint64_t addp(uint64_t a, uint64_t b) { unsigned __int128 s = (unsigned __int128)a + b; if (s >= P) s -= P; return (uint64_t)s; }
/* first row a_k = seed + k + 1, b_k = 2k + 3: every seed is a different valid witness (sub-proof input) */
void widefib_trace_seed(unsigned nbits, unsigned W, uint64_t seed, uint64_t *out) {
    uint64_t N = (uint64_t)1 << nbits;
    for (unsigned k = 0; k < W; ++k) { out[2 * k] = (seed + k + 1) % P; out[2 * k + 1] = 2 * k + 3; }
    for (uint64_t i = 1; i < N; ++i)
        for (unsigned k = 0; k < W; ++k) {
            const uint64_t *p = out + (i - 1) * 2 * W + 2 * k;
            uint64_t *c = out + i * 2 * W + 2 * k;
            c[0] = p[1]; c[1] = addp(p[0], p[1]);
        }
}
void widefib_trace(unsigned nbits, unsigned W, uint64_t *out) { widefib_trace_seed(nbits, W, 0, out); }

/* ---- PoseidonG state machine (BASELINE config 3 and the 2^24 headline) -------------------------------------
 * Inputs of `starkjs/poseidon/poseidong.pil`, semantics of starkjs/poseidon/sm_poseidong.js:
 * buildConstants (:110-133) and execute (:136-280).  31-row blocks: row r of a block holds the state before
 * round r (4 full, 22 partial, 4 full), row 30 the permutation's output; hash0..3 carry the digest through the
 * block.  Blocks [0, n_inputs) hash distinct inputs (block 0: `first`, the reference's own test inputs
 * main_poseidon.js:50-58; block k > 0: splitmix64(seed, 12k + j) mod p), the rest repeats the all-zero-input
 * permutation as the reference pads (:241-276).  Layouts are the .const / .cm files' (polsarray.rs:137-217):
 * row-major little-endian u64, columns in declaration order. */
#include "poseidong_round_constants.h"
#define PG_ROUNDS 30
#define PG_BLOCK 31
static inline uint64_t mulp(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
static inline uint64_t pow7(uint64_t a) { uint64_t a2 = mulp(a, a), a4 = mulp(a2, a2), a3 = mulp(a, a2); return mulp(a3, a4); }
static const uint64_t PG_MCIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static void pg_round(uint64_t st[12], int r) {
    uint64_t t[12];
    for (int i = 0; i < 12; ++i) t[i] = addp(st[i], POSEIDONG_C[12 * r + i]);
    if (r < 4 || r >= 26) { for (int i = 0; i < 12; ++i) t[i] = pow7(t[i]); } else t[0] = pow7(t[0]);
    for (int i = 0; i < 12; ++i) {                                   /* M[i][j] = circ[(j - i) mod 12] + 8 [i = j = 0] */
        unsigned __int128 acc = 0;
        for (int j = 0; j < 12; ++j) acc += (unsigned __int128)(PG_MCIRC[(j - i + 12) % 12] + (i == 0 && j == 0 ? 8 : 0)) * t[j];
        st[i] = (uint64_t)(acc % P);
    }
}
static inline uint64_t pg_splitmix(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    return z % P;
}
/* out: [N][18] = LAST, LATCH, LASTBLOCK, PARTIAL, C[0..11], LINPUT, LOUTPUT */
void poseidong_consts(unsigned nbits, uint64_t *out) {
    uint64_t N = (uint64_t)1 << nbits, max_hashes = N / PG_BLOCK;
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < N; ++i) {
        uint64_t *o = out + i * 18, r = i % PG_BLOCK, ih = i / PG_BLOCK;
        o[0] = (i == N - 1 || r == PG_ROUNDS) ? 1 : 0;
        o[1] = (ih < max_hashes && r == 0) ? 1 : 0;
        o[2] = (r == PG_ROUNDS) ? 1 : 0;
        o[3] = (r >= 4 && r < 26) ? 1 : 0;
        for (int j = 0; j < 12; ++j) o[4 + j] = POSEIDONG_C[12 * r + j];
        o[16] = (i == 0) ? 1 : 0;
        o[17] = (i == N - 1) ? 1 : 0;
    }
}
/* out: [N][19] = in0..7, hashType, cap1..3, hash0..3, result1..3.  Returns 0, or -1 for too many inputs. */
int poseidong_trace(unsigned nbits, uint64_t n_inputs, const uint64_t first[12], uint64_t seed, uint64_t *out) {
    uint64_t N = (uint64_t)1 << nbits, max_hashes = N / PG_BLOCK, n_blocks = (N + PG_BLOCK - 1) / PG_BLOCK;
    if (n_inputs > max_hashes) return -1;                              /* "Not enough Poseidon slots" (:151-153) */
    #pragma omp parallel for schedule(static)
    for (uint64_t b = 0; b < n_blocks; ++b) {
        uint64_t st[PG_BLOCK][12];
        for (int j = 0; j < 12; ++j) st[0][j] = b >= n_inputs ? 0 : b == 0 ? first[j] % P : pg_splitmix(seed, 12 * b + j);
        for (int r = 0; r < PG_ROUNDS; ++r) { for (int j = 0; j < 12; ++j) st[r + 1][j] = st[r][j]; pg_round(st[r + 1], r); }
        for (uint64_t r = 0; r < PG_BLOCK && b * PG_BLOCK + r < N; ++r) {
            uint64_t *o = out + (b * PG_BLOCK + r) * 19;
            for (int j = 0; j < 12; ++j) o[j] = st[r][j];
            for (int j = 0; j < 4; ++j) o[12 + j] = st[PG_ROUNDS][j];
            o[16] = o[17] = o[18] = 0;
        }
    }
    return 0;
}
