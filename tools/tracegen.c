/* Synthetic input generator (not oracle, not product): "wide Fibonacci" execution trace for
 * tools/synth_pil.py -- W independent pairs (a_k, b_k) per row, a' = b, b' = a + b (mod p),
 * row-major [N][2W] little-endian u64, plus splitmix64 columns for the NTT benchmark.
 * Build: gcc -O2 -shared -fPIC tools/tracegen.c -o tools/libtracegen.so */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
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
/* a * b mod p with 2^64 = 2^32 - 1, 2^96 = -1 (a 128-bit `%` through libgcc costs ten times as much) */
static inline uint64_t mulp(uint64_t a, uint64_t b) {
    const unsigned __int128 x = (unsigned __int128)a * b;
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    uint64_t t0 = lo - hh; if (lo < hh) t0 -= 0xFFFFFFFFULL;
    const uint64_t t1 = (hl << 32) - hl;
    uint64_t t2 = t0 + t1; if (t2 < t1) t2 += 0xFFFFFFFFULL;
    return t2 >= P ? t2 - P : t2;
}
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

/* ---- compressor-shaped PLONK circuit (tools/pil/c12_shape.pil; BASELINE config 5's 2^15 and 2^18 STARKs) ---------------
 * One fixed circuit per size (gate coefficients and wiring from `seed`), many witnesses (one per sub-proof, from its primary
 * inputs) -- as in test/recursive_proof_to_snark.sh, where every task proves the same verifier circuit on its own input.
 * Row r: the 8 gate inputs (columns 0,1,3,4,6,7,9,10) read wires drawn from those defined so far, the 4 gate outputs
 * (columns 2,5,8,11: a2 = C3 a0 a1 + C0 a0 + C1 a1 + C4 with C2 = -1, ...) define new ones.  The copy constraints link the
 * cells of a wire in a cycle: S[col][row] = k_col' * w^row' for the next cell (row', col') of the same wire, k_0 = 1,
 * k_i = k^i (helper.rs:16-23, starkinfo_Z.rs:273-423).
 * const: [N][26] = Global.L1, S[0..11], C[0..11], GATE.  cm: [N][12].  wires: [N][8] input wire ids (the circuit).     */
#define C12S_PRIMARY 16
static const int C12S_IN[8] = {0, 1, 3, 4, 6, 7, 9, 10};
static const int C12S_OUT[4] = {2, 5, 8, 11};
static inline uint64_t powp(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = mulp(r, a); a = mulp(a, a); e >>= 1; } return r; }
void c12s_circuit(unsigned nbits, uint64_t seed, uint64_t root_of_unity /* MG.0[nbits] */, uint64_t *consts, uint32_t *wires) {
    const uint64_t N = (uint64_t)1 << nbits, K = 12275445934081160404ULL, n_wires = C12S_PRIMARY + 4 * N;
    uint64_t ks[12]; ks[0] = 1; for (int i = 1; i < 12; ++i) ks[i] = mulp(ks[i - 1], K);
    int64_t *first = (int64_t *)malloc(n_wires * sizeof(int64_t)), *last = (int64_t *)malloc(n_wires * sizeof(int64_t));
    int64_t *sigma = (int64_t *)malloc(12 * N * sizeof(int64_t));       /* cell = 12 * row + col */
    uint64_t *wpow = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint64_t w = 0; w < n_wires; ++w) first[w] = last[w] = -1;
    wpow[0] = 1; for (uint64_t i = 1; i < N; ++i) wpow[i] = mulp(wpow[i - 1], root_of_unity);
    uint64_t ctr = 0;
    for (uint64_t r = 0; r < N; ++r) {
        uint64_t *o = consts + r * 26;
        for (int j = 0; j < 26; ++j) o[j] = 0;
        o[0] = r == 0; o[25] = 1;
        uint64_t *C = o + 13;
        C[0] = pg_splitmix(seed, ctr++); C[1] = pg_splitmix(seed, ctr++); C[3] = pg_splitmix(seed, ctr++); C[4] = pg_splitmix(seed, ctr++); C[2] = P - 1;
        C[6] = pg_splitmix(seed, ctr++); C[7] = pg_splitmix(seed, ctr++); C[9] = pg_splitmix(seed, ctr++); C[10] = pg_splitmix(seed, ctr++); C[8] = P - 1;
        const uint64_t avail = C12S_PRIMARY + 4 * r;
        for (int g = 0; g < 12; ++g) {
            uint64_t w;
            int is_out = -1;
            for (int k = 0; k < 4; ++k) if (C12S_OUT[k] == g) is_out = k;
            if (is_out >= 0) w = avail + is_out;
            else {
                int slot = 0; for (int k = 0; k < 8; ++k) if (C12S_IN[k] == g) slot = k;
                /* row 0 reads the first primary inputs in order (they are the publics); later rows draw recent wires
                 * more often than old ones, as a compiled circuit does */
                if (r == 0) w = slot;
                else { uint64_t x = pg_splitmix(seed ^ 0xC125, ctr++); w = (x & 1) ? avail - 1 - (x >> 1) % (avail < 64 ? avail : 64) : (x >> 1) % avail; }
                wires[r * 8 + slot] = (uint32_t)w;
            }
            const int64_t cell = 12 * (int64_t)r + g;
            if (first[w] < 0) first[w] = cell; else sigma[last[w]] = cell;
            last[w] = cell;
        }
    }
    for (uint64_t w = 0; w < n_wires; ++w) if (first[w] >= 0) sigma[last[w]] = first[w];
    #pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < N; ++r)
        for (int g = 0; g < 12; ++g) { const int64_t t = sigma[12 * r + g]; consts[r * 26 + 1 + g] = mulp(ks[t % 12], wpow[t / 12]); }
    free(first); free(last); free(sigma); free(wpow);
}
void c12s_witness(unsigned nbits, const uint64_t *consts, const uint32_t *wires, const uint64_t primary[C12S_PRIMARY], uint64_t *cm) {
    const uint64_t N = (uint64_t)1 << nbits;
    uint64_t *val = (uint64_t *)malloc((C12S_PRIMARY + 4 * N) * sizeof(uint64_t));
    for (int i = 0; i < C12S_PRIMARY; ++i) val[i] = primary[i] % P;
    for (uint64_t r = 0; r < N; ++r) {
        const uint64_t *C = consts + r * 26 + 13;
        uint64_t *a = cm + r * 12;
        for (int k = 0; k < 8; ++k) a[C12S_IN[k]] = val[wires[r * 8 + k]];
        for (int g = 0; g < 4; ++g) {
            const uint64_t x = a[3 * g], y = a[3 * g + 1], *c = C + (g < 2 ? 0 : 6);
            a[3 * g + 2] = addp(addp(mulp(c[3], mulp(x, y)), mulp(c[0], x)), addp(mulp(c[1], y), c[4]));
            val[C12S_PRIMARY + 4 * r + g] = a[3 * g + 2];
        }
    }
    free(val);
}
/* ---- the join circuit of the aggregation (recursive2 in test/stark_aggregation.sh:80-156): same PIL shape, LINEAR gates in layers ----
 * The real recursive2 witness is what its circom calculator writes, followed by compressor12 exec: PlonkAdd sums + the s_map
 * gather (compressor12_exec.rs:58-103).  This stand-in keeps only what exec can compute, so that a join needs no host walk:
 * every gate is out = C0 x + C1 y (C3 = C4 = 0, C2 = -1), i.e. ONE PlonkAdd, and the inputs of a row in layer l (rows
 * [l 2^layer_bits, (l + 1) 2^layer_bits)) are drawn from the wires of earlier layers only (layer 0: the primary inputs), so the
 * additions form a DAG of N / 2^layer_bits levels.  Witness vector of the exec: w[0] = 1 (circom's constant signal), w[1..16] = the
 * primary inputs, add k = gate (row k / 4, k mod 4) -> w[17 + k].  Same constants layout and wiring format as c12s_circuit. */
void c12l_circuit(unsigned nbits, unsigned layer_bits, uint64_t seed, uint64_t root_of_unity, uint64_t *consts, uint32_t *wires) {
    const uint64_t N = (uint64_t)1 << nbits, K = 12275445934081160404ULL, n_wires = C12S_PRIMARY + 4 * N;
    uint64_t ks[12]; ks[0] = 1; for (int i = 1; i < 12; ++i) ks[i] = mulp(ks[i - 1], K);
    int64_t *first = (int64_t *)malloc(n_wires * sizeof(int64_t)), *last = (int64_t *)malloc(n_wires * sizeof(int64_t));
    int64_t *sigma = (int64_t *)malloc(12 * N * sizeof(int64_t));
    uint64_t *wpow = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (uint64_t w = 0; w < n_wires; ++w) first[w] = last[w] = -1;
    wpow[0] = 1; for (uint64_t i = 1; i < N; ++i) wpow[i] = mulp(wpow[i - 1], root_of_unity);
    uint64_t ctr = 0;
    for (uint64_t r = 0; r < N; ++r) {
        uint64_t *o = consts + r * 26;
        for (int j = 0; j < 26; ++j) o[j] = 0;
        o[0] = r == 0; o[25] = 1;
        uint64_t *C = o + 13;
        C[0] = pg_splitmix(seed, ctr++); C[1] = pg_splitmix(seed, ctr++); C[2] = P - 1;
        C[6] = pg_splitmix(seed, ctr++); C[7] = pg_splitmix(seed, ctr++); C[8] = P - 1;
        const uint64_t avail = C12S_PRIMARY + 4 * ((r >> layer_bits) << layer_bits);   /* wires of earlier layers */
        for (int g = 0; g < 12; ++g) {
            uint64_t w;
            int is_out = -1;
            for (int k = 0; k < 4; ++k) if (C12S_OUT[k] == g) is_out = k;
            if (is_out >= 0) w = C12S_PRIMARY + 4 * r + is_out;
            else {
                int slot = 0; for (int k = 0; k < 8; ++k) if (C12S_IN[k] == g) slot = k;
                if (r == 0) w = slot;                                                /* the publics: a0, a1, a3 of row 0 */
                else { uint64_t x = pg_splitmix(seed ^ 0xC12E, ctr++); w = (x & 1) ? avail - 1 - (x >> 1) % (avail < 4096 ? avail : 4096) : (x >> 1) % avail; }
                wires[r * 8 + slot] = (uint32_t)w;
            }
            const int64_t cell = 12 * (int64_t)r + g;
            if (first[w] < 0) first[w] = cell; else sigma[last[w]] = cell;
            last[w] = cell;
        }
    }
    for (uint64_t w = 0; w < n_wires; ++w) if (first[w] >= 0) sigma[last[w]] = first[w];
    #pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < N; ++r)
        for (int g = 0; g < 12; ++g) { const int64_t t = sigma[12 * r + g]; consts[r * 26 + 1 + g] = mulp(ks[t % 12], wpow[t / 12]); }
    free(first); free(last); free(sigma); free(wpow);
}
/* The circuit's .exec file (compressor12_setup.rs:51-83): [adds_len, s_map_column_len, adds (a, b, raw ca, raw cb), s_map row by row];
 * coefficients as the raw words of an FGL, value * 2^64 mod p (field_gl.rs:503-507).  Returns the text length (without the
 * terminating zero); writes at most cap bytes -- call with cap = 0 for the size. */
uint64_t c12l_exec_text(unsigned nbits, const uint64_t *consts, const uint32_t *wires, char *buf, uint64_t cap) {
    const uint64_t N = (uint64_t)1 << nbits, R64 = 0xFFFFFFFFULL;                    /* 2^64 mod p */
    uint64_t len = 0;
    char tmp[32];
    #define EMIT_U64(v) do { int n_ = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)(v)); if (len + n_ <= cap) memcpy(buf + len, tmp, n_); len += n_; } while (0)
    #define EMIT_CH(c) do { if (len + 1 <= cap) buf[len] = (c); len += 1; } while (0)
    EMIT_CH('['); EMIT_U64(4 * N); EMIT_CH(','); EMIT_U64(N);
    for (uint64_t r = 0; r < N; ++r)
        for (int g = 0; g < 4; ++g) {
            const uint64_t *c = consts + r * 26 + 13 + (g < 2 ? 0 : 6);
            EMIT_CH(','); EMIT_U64(1 + wires[r * 8 + 2 * g]); EMIT_CH(','); EMIT_U64(1 + wires[r * 8 + 2 * g + 1]);
            EMIT_CH(','); EMIT_U64(mulp(c[0], R64)); EMIT_CH(','); EMIT_U64(mulp(c[1], R64));
        }
    for (uint64_t r = 0; r < N; ++r)
        for (int col = 0; col < 12; ++col) {
            uint64_t w;
            if (col % 3 == 2) w = C12S_PRIMARY + 4 * r + col / 3;                      /* a gate output: its add */
            else w = wires[r * 8 + 2 * (col / 3) + col % 3];
            EMIT_CH(','); EMIT_U64(1 + w);
        }
    EMIT_CH(']');
    #undef EMIT_U64
    #undef EMIT_CH
    return len;
}
/* the starkjs Fibonacci circuit of the first STARK of a task (fibonacci.js:8-27): const [N][2] = L1, LLAST; cm [N][2] = l1, l2 */
void fib_consts(unsigned nbits, uint64_t *out) {
    const uint64_t N = (uint64_t)1 << nbits;
    for (uint64_t i = 0; i < N; ++i) { out[2 * i] = i == 0; out[2 * i + 1] = i == N - 1; }
}
void fib_trace(unsigned nbits, uint64_t in0, uint64_t in1, uint64_t *out) {
    const uint64_t N = (uint64_t)1 << nbits;
    out[1] = in0 % P; out[0] = in1 % P;                                   /* l2[0] = input[0], l1[0] = input[1] */
    for (uint64_t i = 1; i < N; ++i) {
        out[2 * i + 1] = out[2 * (i - 1)];
        out[2 * i] = addp(mulp(out[2 * (i - 1) + 1], out[2 * (i - 1) + 1]), mulp(out[2 * (i - 1)], out[2 * (i - 1)]));
    }
}
