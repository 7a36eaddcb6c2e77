/* ORACLE -- TEST INFRASTRUCTURE ONLY (see gl.h header).  CPU restatement of the starky prover
 * primitives of eigen-zkvm; every function cites the reference file:line it follows
 * (paths under /root/reference/starky/src unless stated).
 *
 * Parity pinning: tests/test_oracle_kat.py checks this file against every known-answer vector the
 * reference's own unit tests hold for the path.
 */
#include "gl.h"
#include "poseidon_gl_constants.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------- */
/* Field helpers exported for the python tests                                                  */
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t orc_gl_mul_slow(uint64_t a, uint64_t b) { return gl_mul_slow(a, b); }
uint64_t orc_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t orc_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t orc_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
uint64_t orc_gl_root(unsigned k) { return gl_root(k); }
void orc_f3_mul(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]) {
    f3_t x, y; memcpy(x.v, a, 24); memcpy(y.v, b, 24);
    f3_t r = f3_mul(x, y); memcpy(out, r.v, 24);
}
void orc_f3_inv(const uint64_t a[3], uint64_t out[3]) {
    f3_t x; memcpy(x.v, a, 24);
    f3_t r = f3_inv(x); memcpy(out, r.v, 24);
}
void orc_f3_pow(const uint64_t a[3], uint64_t e, uint64_t out[3]) {
    f3_t x; memcpy(x.v, a, 24);
    f3_t r = f3_pow(x, e); memcpy(out, r.v, 24);
}

/* ------------------------------------------------------------------------------------------- */
/* fft_p.rs:14-32 BR(): bit reversal of the low `bits` bits                                       */
/* worker threads the parallel loops of this library use (bench.py reports it as cpu_baseline.cores) */
int orc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* cap the team at the CPUs this process may really use (tests/oracle_lib.py reads the cgroup quota: a container that sees
 * 256 logical CPUs but is allowed 16 runs slower with 128 threads than with 16) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

uint32_t orc_bitrev(uint32_t x, unsigned bits) {
    uint32_t r = x; /* full 32-bit reversal, then >> (32 - bits): bits of x above `bits` spill in, as in the reference */
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    r = ((r >> 2) & 0x33333333u) | ((r & 0x33333333u) << 2);
    r = ((r >> 4) & 0x0F0F0F0Fu) | ((r & 0x0F0F0F0Fu) << 4);
    r = ((r >> 8) & 0x00FF00FFu) | ((r & 0x00FF00FFu) << 8);
    r = (r >> 16) | (r << 16);
    return bits ? r >> (32 - bits) : 0;
}

/* fft.rs:39-72: textbook radix-2 DIT NTT, natural order in and out, stage-s twiddle MG.0[s].
 * One column of `stride`-spaced elements transformed in place through a scratch buffer.       */
static void ntt_column(uint64_t *buf, size_t n, unsigned bits, const uint64_t *roots /* n/2 powers of MG[bits] */) {
    for (size_t i = 0; i < n; ++i) {
        size_t r = orc_bitrev((uint32_t)i, bits);
        if (r > i) { uint64_t t = buf[i]; buf[i] = buf[r]; buf[r] = t; }
    }
    for (unsigned s = 1; s <= bits; ++s) {
        size_t m = (size_t)1 << s, md2 = m >> 1, step = n >> s;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < md2; ++j) {
                uint64_t t = gl_mul(roots[j * step], buf[k + j + md2]);
                uint64_t u = buf[k + j];
                buf[k + j] = gl_add(u, t);
                buf[k + j + md2] = gl_sub(u, t);
            }
    }
}

/* the same transform with every stage's n/2 butterflies spread over the OpenMP team: used when the matrix has
 * fewer columns than threads (the reference parallelises inside a transform too, fft_p.rs:214-239), so that
 * the timed CPU baseline of a single-column NTT is not a one-core number */
static void ntt_column_par(uint64_t *buf, size_t n, unsigned bits, const uint64_t *roots) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        size_t r = orc_bitrev((uint32_t)i, bits);
        if (r > i) { uint64_t t = buf[i]; buf[i] = buf[r]; buf[r] = t; }
    }
    for (unsigned s = 1; s <= bits; ++s) {
        size_t m = (size_t)1 << s, md2 = m >> 1, step = n >> s;
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n / 2; ++i) {
            size_t k = (i >> (s - 1)) << s, j = i & (md2 - 1);
            uint64_t t = gl_mul(roots[j * step], buf[k + j + md2]);
            uint64_t u = buf[k + j];
            buf[k + j] = gl_add(u, t);
            buf[k + j + md2] = gl_sub(u, t);
        }
    }
}

static uint64_t *make_roots(unsigned bits) {
    size_t half = bits ? ((size_t)1 << (bits - 1)) : 1;
    uint64_t *r = (uint64_t *)malloc(half * sizeof(uint64_t));
    uint64_t w = gl_root(bits), c = 1;
    for (size_t i = 0; i < half; ++i) { r[i] = c; c = gl_mul(c, w); }
    return r;
}

int orc_ntt(const uint64_t *src, uint64_t *dst, uint32_t n_pols, uint32_t nbits, int inverse);
/* ---- cpu_baseline of bench.py's NTT line: one column, every host thread busy, structured as the reference's parallel
 * transform is (fft_p.rs:174-239: in-cache block transforms separated by full-matrix transposes; here the blocks are the
 * rows of the sqrt(N) x sqrt(N) view -- the textbook radix-2 above walks the whole 128 MiB column once per stage and says
 * nothing about what a CPU can do).  Same values as orc_ntt (tests/test_oracle_kat.py compares them). */
static void transpose_par(const uint64_t *in, uint64_t *out, size_t rows, size_t cols) {   /* out[c][r] = in[r][c] */
    const size_t T = 32;
    #pragma omp parallel for collapse(2) schedule(static)
    for (size_t r0 = 0; r0 < rows; r0 += T)
        for (size_t c0 = 0; c0 < cols; c0 += T)
            for (size_t r = r0; r < r0 + T && r < rows; ++r)
                for (size_t c = c0; c < c0 + T && c < cols; ++c) out[c * rows + r] = in[r * cols + c];
}
int orc_ntt_blocked(const uint64_t *src, uint64_t *dst, uint32_t nbits, int inverse) {
    if (nbits < 4) return orc_ntt(src, dst, 1, nbits, inverse);
    const unsigned b1 = nbits / 2, b2 = nbits - b1;
    const size_t n = (size_t)1 << nbits, n1 = (size_t)1 << b1, n2 = (size_t)1 << b2;
    uint64_t *t = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *r1 = make_roots(b1), *r2 = make_roots(b2);
    const uint64_t w = gl_root(nbits);
    transpose_par(src, t, n1, n2);                                    /* t[i2][i1] */
    #pragma omp parallel for schedule(static)
    for (size_t i2 = 0; i2 < n2; ++i2) {
        uint64_t *row = t + i2 * n1;
        ntt_column(row, n1, b1, r1);                                  /* -> t[i2][k1] */
        uint64_t wi = gl_pow(w, i2), f = 1;                           /* twiddle w^(i2 k1) */
        for (size_t k1 = 0; k1 < n1; ++k1) { row[k1] = gl_mul(row[k1], f); f = gl_mul(f, wi); }
    }
    transpose_par(t, dst, n2, n1);                                    /* dst[k1][i2] */
    #pragma omp parallel for schedule(static)
    for (size_t k1 = 0; k1 < n1; ++k1) ntt_column(dst + k1 * n2, n2, b2, r2);   /* -> dst[k1][k2] */
    transpose_par(dst, t, n1, n2);                                    /* t[k2][k1] = X[k1 + n1 k2] */
    if (!inverse) memcpy(dst, t, n * sizeof(uint64_t));
    else {                                                            /* fft.rs:74-83 */
        const uint64_t n_inv = gl_inv(gl_red((uint64_t)n));
        dst[0] = gl_mul(t[0], n_inv);
        #pragma omp parallel for schedule(static)
        for (size_t i = 1; i < n; ++i) dst[i] = gl_mul(t[n - i], n_inv);
    }
    free(t); free(r1); free(r2);
    return 0;
}

/* fft_p.rs:242-253 fft / ifft: batched NTT over a row-major [1<<nbits][n_pols] matrix.
 * Inverse = forward NTT, then res[0]=q[0]/n, res[i]=q[n-i]/n (fft.rs:74-83; fused in
 * fft_p.rs:124-142 as the input permutation (n-BR(i))%n and the 1/n factor).                  */
int orc_ntt(const uint64_t *src, uint64_t *dst, uint32_t n_pols, uint32_t nbits, int inverse) {
    size_t n = (size_t)1 << nbits;
    uint64_t *roots = make_roots(nbits);
    uint64_t n_inv = gl_inv(gl_red((uint64_t)n));
    if ((int)n_pols < orc_threads() && nbits >= 14) {            /* few columns: parallelism inside each transform */
        uint64_t *col = (uint64_t *)malloc(n * sizeof(uint64_t));
        for (uint32_t c = 0; c < n_pols; ++c) {
            #pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; ++i) col[i] = src[i * n_pols + c];
            ntt_column_par(col, n, nbits, roots);
            if (!inverse) {
                #pragma omp parallel for schedule(static)
                for (size_t i = 0; i < n; ++i) dst[i * n_pols + c] = col[i];
            } else {
                dst[c] = gl_mul(col[0], n_inv);
                #pragma omp parallel for schedule(static)
                for (size_t i = 1; i < n; ++i) dst[i * n_pols + c] = gl_mul(col[n - i], n_inv);
            }
        }
        free(col); free(roots);
        return 0;
    }
    #pragma omp parallel
    {
        uint64_t *col = (uint64_t *)malloc(n * sizeof(uint64_t));
        #pragma omp for schedule(static)
        for (uint32_t c = 0; c < n_pols; ++c) {
            for (size_t i = 0; i < n; ++i) col[i] = src[i * n_pols + c];
            ntt_column(col, n, nbits, roots);
            if (!inverse) {
                for (size_t i = 0; i < n; ++i) dst[i * n_pols + c] = col[i];
            } else {
                dst[c] = gl_mul(col[0], n_inv);
                for (size_t i = 1; i < n; ++i) dst[i * n_pols + c] = gl_mul(col[n - i], n_inv);
            }
        }
        free(col);
    }
    free(roots);
    return 0;
}

/* fft_p.rs:255-355 interpolate (production) == polutils.rs:25-33 extend_pol (definition):
 * ext = NTT_Nx( [ iNTT_N(col)[i] * 49^i  for i < N ] || 0^(Nx-N) ).                            */
int orc_lde(const uint64_t *src, uint32_t n_pols, uint32_t nbits, uint64_t *dst, uint32_t nbits_ext) {
    if (n_pols == 0) return 0; /* fft_p.rs:262-264 */
    size_t n = (size_t)1 << nbits, nx = (size_t)1 << nbits_ext;
    uint64_t *roots = make_roots(nbits), *rootsx = make_roots(nbits_ext);
    uint64_t n_inv = gl_inv(gl_red((uint64_t)n));
    #pragma omp parallel
    {
        uint64_t *col = (uint64_t *)malloc(nx * sizeof(uint64_t));
        #pragma omp for schedule(static)
        for (uint32_t c = 0; c < n_pols; ++c) {
            for (size_t i = 0; i < n; ++i) col[i] = src[i * n_pols + c];
            ntt_column(col, n, nbits, roots);
            /* inverse ordering + 1/N + shift^i  (fft_p.rs:144-172, fft_worker.rs:4-22) */
            uint64_t w = n_inv;
            for (size_t i = 0; i <= n / 2 && i < n; ++i) {
                size_t j = (n - i) % n;
                uint64_t a = col[j], b = col[i];
                /* coefficient i = q[(n-i)%n]; swap pairs so the permutation is applied in place */
                col[i] = a; col[j] = b;
            }
            for (size_t i = 0; i < n; ++i) { col[i] = gl_mul(col[i], w); w = gl_mul(w, GL_SHIFT); }
            for (size_t i = n; i < nx; ++i) col[i] = 0;
            ntt_column(col, nx, nbits_ext, rootsx);
            for (size_t i = 0; i < nx; ++i) dst[i * n_pols + c] = col[i];
        }
        free(col);
    }
    free(roots); free(rootsx);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* poseidon_opt.rs:80-200 hash_inner.  in[8] || cap[4] -> first n_out (<=12) words of the state. */
static inline uint64_t pow7(uint64_t x) { /* poseidon_opt.rs:68-74 */
    uint64_t x2 = gl_mul(x, x), x3 = gl_mul(x2, x), x6 = gl_mul(x3, x3);
    return gl_mul(x6, x);
}
static void mat12(const uint64_t *Mx, uint64_t st[12]) { /* out[i] = sum_j M[j][i]*st[j]  (:111-119) */
    uint64_t t[12];
    for (int i = 0; i < 12; ++i) {
        uint64_t acc = 0;
        for (int j = 0; j < 12; ++j) acc = gl_add(acc, gl_mul(Mx[j * 12 + i], st[j]));
        t[i] = acc;
    }
    memcpy(st, t, sizeof t);
}
void orc_poseidon(const uint64_t in[8], const uint64_t cap[4], uint64_t *out, int n_out) {
    const uint64_t *C = ORC_POSEIDON_C, *M = ORC_POSEIDON_M, *P = ORC_POSEIDON_P, *S = ORC_POSEIDON_S;
    uint64_t st[12];
    for (int i = 0; i < 8; ++i) st[i] = gl_red(in[i]);
    for (int i = 0; i < 4; ++i) st[8 + i] = gl_red(cap[i]);
    for (int i = 0; i < 12; ++i) st[i] = gl_add(st[i], C[i]);                       /* :101 */
    for (int r = 0; r < 3; ++r) {                                                   /* :104-124 */
        for (int i = 0; i < 12; ++i) st[i] = gl_add(pow7(st[i]), C[(r + 1) * 12 + i]);
        mat12(M, st);
    }
    for (int i = 0; i < 12; ++i) st[i] = gl_add(pow7(st[i]), C[48 + i]);            /* :126-129 */
    mat12(P, st);                                                                    /* :131-143 */
    for (int r = 0; r < 22; ++r) {                                                   /* :145-164 */
        st[0] = gl_add(pow7(st[0]), C[60 + r]);
        uint64_t s0 = 0;
        for (int j = 0; j < 12; ++j) s0 = gl_add(s0, gl_mul(S[23 * r + j], st[j]));
        for (int k = 1; k < 12; ++k) st[k] = gl_add(st[k], gl_mul(S[23 * r + 11 + k], st[0]));
        st[0] = s0;
    }
    for (int r = 0; r < 3; ++r) {                                                    /* :166-186 */
        for (int i = 0; i < 12; ++i) st[i] = gl_add(pow7(st[i]), C[82 + 12 * r + i]);
        mat12(M, st);
    }
    for (int i = 0; i < 12; ++i) st[i] = pow7(st[i]);                                /* :188-198 */
    mat12(M, st);
    for (int i = 0; i < n_out; ++i) out[i] = st[i];
}

/* linearhash.rs:119-145 _hash: sponge, rate 8, capacity carried, zero-padded tail, <=4 -> identity */
static void lh_sponge(const uint64_t *v, size_t n, uint64_t st[4]) {
    memset(st, 0, 32);
    if (n <= 4) { for (size_t i = 0; i < n; ++i) st[i] = v[i]; return; }
    uint64_t blk[8];
    size_t i = 0;
    while (i < n) {
        size_t k = n - i < 8 ? n - i : 8;
        memcpy(blk, v + i, k * 8);
        for (size_t j = k; j < 8; ++j) blk[j] = 0;
        uint64_t t[4];
        orc_poseidon(blk, st, t, 4);
        memcpy(st, t, 32);
        i += k;
    }
}
/* linearhash.rs:79-110 hash (batch_size 0 => bs = max(8, ceil(len/4))) */
void orc_linearhash(const uint64_t *v, size_t n, uint64_t out[4]) {
    size_t bs = (n + 3) / 4; if (bs < 8) bs = 8;
    memset(out, 0, 32);
    if (n <= 4) { for (size_t i = 0; i < n; ++i) out[i] = v[i]; return; }
    size_t hsz = (n + bs - 1) / bs;
    uint64_t hashes[16];
    for (size_t b = 0; b < hsz; ++b) {
        size_t len = n - b * bs < bs ? n - b * bs : bs;
        lh_sponge(v + b * bs, len, hashes + 4 * b);
    }
    if (hsz * 4 <= 4) memcpy(out, hashes, 32);
    else lh_sponge(hashes, hsz * 4, out);
}

/* merklehash.rs:47-61 get_n_nodes */
uint64_t orc_merkle_n_nodes(uint64_t n_) {
    uint64_t n = n_, next_n = (n - 1) / 2 + 1, acc = next_n * 2;
    while (n > 1) {
        n = next_n; next_n = (n - 1) / 2 + 1;
        if (n > 1) acc += next_n * 2; else acc += 1;
    }
    return acc;
}
/* merklehash.rs:293-346 merkelize: leaves = linearhash(row); parent = Poseidon(L||R, cap 0);
 * level L+1 starts at p_in + 2*ceil(n_L/2); absent sibling = zero digest (nodes zero-initialised). */
int orc_merkelize(const uint64_t *buff, uint32_t width, uint64_t height, uint64_t *nodes /* n_nodes*4, zeroed here */) {
    uint64_t nn = orc_merkle_n_nodes(height);
    memset(nodes, 0, nn * 32);
    #pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < height; ++r) orc_linearhash(buff + r * width, width, nodes + 4 * r);
    uint64_t n64 = height, next = (n64 - 1) / 2 + 1, p_in = 0, p_out = next * 2;
    static const uint64_t zero4[4] = {0, 0, 0, 0};
    while (n64 > 1) {
        #pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < next; ++i) {
            uint64_t t[4];
            orc_poseidon(nodes + 4 * (p_in + 2 * i), zero4, t, 4); /* in[8] = L||R contiguous */
            memcpy(nodes + 4 * (p_out + i), t, 32);
        }
        n64 = next; next = (n64 - 1) / 2 + 1; p_in = p_out; p_out = p_in + next * 2;
    }
    return 0;
}
/* merklehash.rs:64-76 merkle_gen_merkle_proof -> path[depth][4]; returns depth */
int orc_merkle_proof(const uint64_t *nodes, uint64_t height, uint64_t idx, uint64_t *path) {
    uint64_t n = height, off = 0; int d = 0;
    while (n > 1) {
        memcpy(path + 4 * d, nodes + 4 * (off + (idx ^ 1)), 32);
        uint64_t next = (n - 1) / 2 + 1;
        off += next * 2; n = next; idx >>= 1; ++d;
    }
    return d;
}
/* merklehash.rs:150-172 calculate_root_from_group_proof */
void orc_merkle_root_from_proof(const uint64_t *row, uint32_t width, const uint64_t *path, int depth, uint64_t idx, uint64_t root[4]) {
    uint64_t cur[4]; orc_linearhash(row, width, cur);
    static const uint64_t zero4[4] = {0, 0, 0, 0};
    for (int d = 0; d < depth; ++d) {
        uint64_t in[8], t[4];
        if (idx & 1) { memcpy(in, path + 4 * d, 32); memcpy(in + 4, cur, 32); }
        else         { memcpy(in, cur, 32); memcpy(in + 4, path + 4 * d, 32); }
        orc_poseidon(in, zero4, t, 4);
        memcpy(cur, t, 32); idx >>= 1;
    }
    memcpy(root, cur, 32);
}

/* ------------------------------------------------------------------------------------------- */
/* transcript.rs:8-103 TranscriptGL                                                              */
typedef struct { uint64_t state[4]; uint64_t pending[8]; int n_pending; uint64_t out[12]; int out_pos, n_out; } orc_transcript;
void orc_tr_init(orc_transcript *t) { memset(t, 0, sizeof *t); }
static void tr_update(orc_transcript *t) { /* :15-24 */
    for (int i = t->n_pending; i < 8; ++i) t->pending[i] = 0;
    orc_poseidon(t->pending, t->state, t->out, 12);
    t->out_pos = 0; t->n_out = 12; t->n_pending = 0;
    memcpy(t->state, t->out, 32);
}
void orc_tr_put(orc_transcript *t, const uint64_t *v, size_t n) { /* :25-33, :64-71 */
    for (size_t i = 0; i < n; ++i) {
        t->n_out = 0; t->out_pos = 0;
        t->pending[t->n_pending++] = v[i];
        if (t->n_pending == 8) tr_update(t);
    }
}
uint64_t orc_tr_get1(orc_transcript *t) { /* :54-62 */
    if (t->out_pos >= t->n_out) tr_update(t);
    return t->out[t->out_pos++];
}
void orc_tr_get_field(orc_transcript *t, uint64_t out[3]) { /* :47-52 */
    out[0] = orc_tr_get1(t); out[1] = orc_tr_get1(t); out[2] = orc_tr_get1(t);
}
void orc_tr_get_permutations(orc_transcript *t, uint32_t n, uint32_t nbits, uint64_t *res) { /* :73-102 */
    uint64_t total = (uint64_t)n * nbits, n_fields = (total - 1) / 63 + 1;
    uint64_t *f = (uint64_t *)malloc(n_fields * 8);
    for (uint64_t i = 0; i < n_fields; ++i) f[i] = orc_tr_get1(t);
    uint64_t cur_field = 0; unsigned cur_bit = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t a = 0;
        for (uint32_t j = 0; j < nbits; ++j) {
            if ((f[cur_field] >> cur_bit) & 1) a += (uint64_t)1 << j;
            if (++cur_bit == 63) { cur_bit = 0; ++cur_field; }
        }
        res[i] = a;
    }
    free(f);
}
size_t orc_tr_sizeof(void) { return sizeof(orc_transcript); }
