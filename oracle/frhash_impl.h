/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Field-generic body of the scalar-field hashing restatement, included by
 * bn128_hash.c and bls12381_hash.c after they define FH_X(name) (export prefix), FH_RMOD, FH_R2, FH_RINV (modulus,
 * 2^512 mod r, -r^-1 mod 2^64 as 4 x u64 initialisers), FH_NRP (partial rounds of t = 2..17) and FH_OUT_IDX (the
 * state word Poseidon::hash returns).  See bn128_hash.c for the reference files this follows. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;

static const fr_t RMOD = {FH_RMOD};
static const fr_t R2 = {FH_R2};   /* 2^512 mod r (linearhash_*.rs:80-85) */
#define RINV FH_RINV            /* -r^-1 mod 2^64 */

static int fr_geq(const fr_t *a, const fr_t *b) {
    for (int i = 3; i >= 0; --i) { if (a->l[i] > b->l[i]) return 1; if (a->l[i] < b->l[i]) return 0; }
    return 1;
}
static void fr_sub_nored(fr_t *r, const fr_t *a, const fr_t *b) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static fr_t fr_add(fr_t a, fr_t b) {
    fr_t r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || fr_geq(&r, &RMOD)) fr_sub_nored(&r, &r, &RMOD);
    return r;
}
static fr_t fr_mul(fr_t a, fr_t b) { /* Montgomery CIOS: a*b/2^256 mod r; accepts any 256-bit a (result < r if b < r) */
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * RINV;
        c = ((u128)m * RMOD.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * RMOD.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fr_t r = {{t[0], t[1], t[2], t[3]}};
    while (t[4] || fr_geq(&r, &RMOD)) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)r.l[i] - RMOD.l[i] - br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; } t[4] -= (uint64_t)br; }
    return r;
}
static fr_t fr_zero(void) { fr_t z; memset(&z, 0, sizeof z); return z; }
static fr_t fr_to_mont(fr_t a) { return fr_mul(a, R2); }            /* Fr::from_repr; also the "x > MODULUS" branch of to_bn128_mont */
static fr_t fr_from_mont(fr_t a) { fr_t one = {{1, 0, 0, 0}}; return fr_mul(a, one); }   /* into_repr */
static void pow5(fr_t *x) { fr_t a = *x; *x = fr_mul(*x, *x); *x = fr_mul(*x, *x); *x = fr_mul(*x, a); }  /* poseidon_bn128_opt.rs:88-94 */

/* ---- parameter tables (Montgomery form) ------------------------------------------------------------- */
typedef struct { uint32_t t, n_c, n_s, n_rp; fr_t *c, *m, *p, *s; } params_t;
static params_t PRM[16];
static int g_loaded = 0;
static const uint32_t NRP[16] = {FH_NRP};

int FH_X(load_constants)(const char *path) {
    if (g_loaded) return 0;
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char magic[4]; uint32_t nt;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "PBN1", 4) || fread(&nt, 4, 1, f) != 1 || nt != 16) { fclose(f); return -2; }
    for (int k = 0; k < 16; ++k) {
        uint32_t h[3];
        if (fread(h, 4, 3, f) != 3 || h[0] != (uint32_t)k + 2) { fclose(f); return -3; }
        params_t *P = &PRM[k];
        P->t = h[0]; P->n_c = h[1]; P->n_s = h[2]; P->n_rp = NRP[k];
        size_t tt = (size_t)P->t * P->t, n = P->n_c + 2 * tt + P->n_s;
        fr_t *buf = (fr_t *)malloc(n * sizeof(fr_t));
        if (fread(buf, 32, n, f) != n) { fclose(f); return -4; }
        for (size_t i = 0; i < n; ++i) buf[i] = fr_to_mont(buf[i]);
        P->c = buf; P->m = buf + P->n_c; P->p = P->m + tt; P->s = P->p + tt;
    }
    fclose(f);
    g_loaded = 1;
    return 0;
}

/* poseidon_bn128_opt.rs:98-224 hash_inner; operands and results in Montgomery form */
static int poseidon(const fr_t *inp, uint32_t n_in, fr_t init, uint32_t n_out, fr_t *out) {
    if (!g_loaded || n_in == 0 || n_in > 16) return -1;
    const uint32_t t = n_in + 1;
    if (n_out > t) return -1;
    const params_t *P = &PRM[t - 2];
    const uint32_t nrp = P->n_rp;
    fr_t st[17], tmp[17];
    st[0] = init;
    for (uint32_t i = 0; i < n_in; ++i) st[i + 1] = inp[i];
    for (uint32_t i = 0; i < t; ++i) st[i] = fr_add(st[i], P->c[i]);
#define MATMUL(MAT)                                                                                   \
    do {                                                                                              \
        for (uint32_t i = 0; i < t; ++i) {                                                            \
            fr_t acc = fr_zero();                                                                     \
            for (uint32_t j = 0; j < t; ++j) acc = fr_add(acc, fr_mul((MAT)[j * t + i], st[j]));      \
            tmp[i] = acc;                                                                             \
        }                                                                                             \
        memcpy(st, tmp, t * sizeof(fr_t));                                                            \
    } while (0)
    for (uint32_t r = 0; r < 3; ++r) {
        for (uint32_t i = 0; i < t; ++i) { pow5(&st[i]); st[i] = fr_add(st[i], P->c[(r + 1) * t + i]); }
        MATMUL(P->m);
    }
    for (uint32_t i = 0; i < t; ++i) { pow5(&st[i]); st[i] = fr_add(st[i], P->c[4 * t + i]); }
    MATMUL(P->p);
    for (uint32_t r = 0; r < nrp; ++r) {
        pow5(&st[0]);
        st[0] = fr_add(st[0], P->c[5 * t + r]);
        const fr_t *S = P->s + (size_t)(2 * t - 1) * r;
        fr_t s0 = fr_zero();
        for (uint32_t j = 0; j < t; ++j) s0 = fr_add(s0, fr_mul(S[j], st[j]));
        for (uint32_t k = 1; k < t; ++k) st[k] = fr_add(st[k], fr_mul(S[t + k - 1], st[0]));
        st[0] = s0;
    }
    for (uint32_t r = 0; r < 3; ++r) {
        for (uint32_t i = 0; i < t; ++i) { pow5(&st[i]); st[i] = fr_add(st[i], P->c[5 * t + nrp + r * t + i]); }
        MATMUL(P->m);
    }
    for (uint32_t i = 0; i < t; ++i) pow5(&st[i]);
    MATMUL(P->m);
#undef MATMUL
    memcpy(out, st, n_out * sizeof(fr_t));
    return 0;
}

/* Poseidon::hash (poseidon_*_opt.rs:80-86 / :94-103): BN128 returns state[0], BLS12-381 state[1] (Neptune's choice) */
static int hash1(const fr_t *inp, uint32_t n_in, fr_t init, fr_t *out) {
    fr_t o[2];
    int rc = poseidon(inp, n_in, init, FH_OUT_IDX + 1, o);
    if (rc == 0) *out = o[FH_OUT_IDX];
    return rc;
}

/* ---- exported: field helpers, Poseidon ---------------------------------------------------------------- */
void FH_X(fr_to_mont)(const uint64_t a[4], uint64_t r[4]) { fr_t x; memcpy(&x, a, 32); x = fr_to_mont(x); memcpy(r, &x, 32); }
void FH_X(fr_from_mont)(const uint64_t a[4], uint64_t r[4]) { fr_t x; memcpy(&x, a, 32); x = fr_from_mont(x); memcpy(r, &x, 32); }
/* inp / init / out: raw (Montgomery) limbs, as digests carry them */
int FH_X(poseidon)(const uint64_t *inp, uint32_t n_in, const uint64_t init[4], uint32_t n_out, uint64_t *out) {
    fr_t in[16], o[17], is;
    if (n_in > 16) return -1;
    memcpy(in, inp, (size_t)n_in * 32); memcpy(&is, init, 32);
    int rc = poseidon(in, n_in, is, n_out, o);
    if (rc == 0) memcpy(out, o, (size_t)n_out * 32);
    return rc;
}

/* Poseidon::hash: the one-element form used by the trees */
int FH_X(hash)(const uint64_t *inp, uint32_t n_in, const uint64_t init[4], uint64_t out[4]) {
    fr_t in[16], o, is;
    if (n_in > 16) return -1;
    memcpy(in, inp, (size_t)n_in * 32); memcpy(&is, init, 32);
    int rc = hash1(in, n_in, is, &o);
    if (rc == 0) memcpy(out, &o, 32);
    return rc;
}

/* ---- LinearHashBN128 --------------------------------------------------------------------------------- */
/* digest.rs:162-175 to_bn128: e0 + e1*2^64 + e2*2^128 + e3*2^192 -> Montgomery (from_repr: must be < r) */
static fr_t words_to_fr(const uint64_t *e, size_t n) {
    fr_t x = fr_zero();
    for (size_t i = 0; i < n && i < 4; ++i) x.l[i] = e[i];
    return fr_to_mont(x);
}
/* linearhash_bn128.rs:105-131 hash_element_array -> digest (raw limbs) */
int FH_X(hash_element_array)(const uint64_t *vals, uint64_t n, uint64_t out[4]) {
    if (n <= 4) {   /* to_bn128_mont (:70-91): Montgomery form of the 256-bit integer, reduced when >= r */
        fr_t d = words_to_fr(vals, (size_t)n);
        memcpy(out, &d, 32);
        return 0;
    }
    const size_t nb = (size_t)(n - 1) / 3 + 1;
    fr_t *buf = (fr_t *)malloc(nb * sizeof(fr_t));
    for (size_t k = 0; k < nb; ++k) { size_t len = n - 3 * k < 3 ? (size_t)(n - 3 * k) : 3; buf[k] = words_to_fr(vals + 3 * k, len); }
    fr_t digest = fr_zero();
    int rc = 0;
    for (size_t i = 0; i < nb && rc == 0; i += 16) {
        uint32_t sz = nb - i < 16 ? (uint32_t)(nb - i) : 16;
        rc = hash1(buf + i, sz, digest, &digest);
    }
    free(buf);
    memcpy(out, &digest, 32);
    return rc;
}
/* linearhash_bn128.rs:23-67 hash_element_matrix over the concatenated columns -> Fr (raw limbs) */
int FH_X(hash_element_matrix)(const uint64_t *vals, uint64_t n, uint64_t out[4]) {
    const size_t nb = n ? (size_t)(n - 1) / 3 + 1 : 0;
    fr_t st = fr_zero();
    if (nb == 0) { memcpy(out, &st, 32); return 0; }
    fr_t *v3 = (fr_t *)malloc(nb * sizeof(fr_t));
    for (size_t k = 0; k < nb; ++k) { size_t len = n - 3 * k < 3 ? (size_t)(n - 3 * k) : 3; v3[k] = words_to_fr(vals + 3 * k, len); }  /* e0 + e1*2^64 + e2*2^128 */
    int rc = 0;
    if (nb == 1) st = v3[0];
    else
        for (size_t i = 0; i < nb && rc == 0; i += 16) {
            uint32_t sz = nb - i < 16 ? (uint32_t)(nb - i) : 16;
            rc = hash1(v3 + i, sz, st, &st);
        }
    free(v3);
    memcpy(out, &st, 32);
    return rc;
}

/* ---- MerkleTreeBN128 ---------------------------------------------------------------------------------- */
uint64_t FH_X(merkle_n_nodes)(uint64_t n_) {  /* merklehash_bn128.rs:26-39 */
    uint64_t n = n_, next_n = (n - 1) / 16 + 1, acc = next_n * 16;
    while (n > 1) {
        n = next_n; next_n = (n - 1) / 16 + 1;
        if (n > 1) acc += next_n * 16; else acc += 1;
    }
    return acc;
}
/* merklehash_bn128.rs:196-239: nodes[n_nodes][4] raw limbs, zero padded */
int FH_X(merkelize)(const uint64_t *rows, uint32_t width, uint64_t height, uint64_t *nodes) {
    const uint64_t nn = FH_X(merkle_n_nodes)(height);
    memset(nodes, 0, nn * 32);
    int rc = 0;
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < height; ++i) {
        int r = FH_X(hash_element_array)(rows + i * width, width, nodes + 4 * i);
        if (r) rc = r;
    }
    if (rc) return rc;
    uint64_t n = height, next = (n - 1) / 16 + 1, p_in = 0, p_out = next * 16;
    while (n > 1) {
        #pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < next; ++i) {   /* hash_node (linearhash_bn128.rs:93-103): 16 digests, init 0 */
            fr_t in[16], o;
            memcpy(in, nodes + 4 * (p_in + 16 * i), 16 * 32);
            if (hash1(in, 16, fr_zero(), &o)) rc = -1;
            memcpy(nodes + 4 * (p_out + i), &o, 32);
        }
        n = next; next = (n - 1) / 16 + 1; p_in = p_out; p_out = p_in + next * 16;
    }
    return rc;
}
uint32_t FH_X(merkle_depth)(uint64_t height) { uint32_t d = 0; uint64_t n = height; while (n > 1) { n = (n - 1) / 16 + 1; ++d; } return d; }
/* merklehash_bn128.rs:86-106 merkle_gen_merkle_proof: per level the 16 nodes of idx's group; path[depth][16][4] */
void FH_X(merkle_proof)(const uint64_t *nodes, uint64_t height, uint64_t idx, uint64_t *path) {
    uint64_t n = height, off = 0, id = idx; uint32_t d = 0;
    while (n > 1) {
        const uint64_t si = id & ~(uint64_t)15;
        memcpy(path + (size_t)d * 64, nodes + 4 * (off + si), 16 * 32);
        const uint64_t next = (n - 1) / 16 + 1;
        off += next * 16; n = next; id >>= 4; ++d;
    }
}
/* merklehash_bn128.rs:108-138: the root a group proof leads to (the leaf value itself is not re-checked against
 * its slot, exactly as in the reference) */
int FH_X(merkle_root_from_proof)(const uint64_t *path, uint32_t depth, const uint64_t leaf[4], uint64_t out[4]) {
    fr_t v; memcpy(&v, leaf, 32);
    for (uint32_t d = 0; d < depth; ++d) {
        fr_t in[16];
        memcpy(in, path + (size_t)d * 64, 16 * 32);
        if (hash1(in, 16, fr_zero(), &v)) return -1;
    }
    memcpy(out, &v, 32);
    return 0;
}

/* ---- TranscriptBN128 (transcript_bn128.rs:14-132) ---------------------------------------------------- */
typedef struct { fr_t state; fr_t pending[16]; uint32_t n_pending; fr_t out[17]; uint32_t out_pos, n_out; uint64_t out3[3]; uint32_t out3_pos, n_out3; } tr_t;
void *FH_X(tr_new)(void) { return calloc(1, sizeof(tr_t)); }
void FH_X(tr_free)(void *p) { free(p); }
static int tr_update(tr_t *t) {  /* :22-31 */
    while (t->n_pending < 16) t->pending[t->n_pending++] = fr_zero();
    if (poseidon(t->pending, 16, t->state, 17, t->out)) return -1;
    t->out_pos = 0; t->n_out = 17; t->n_out3 = 0; t->out3_pos = 0; t->n_pending = 0;
    t->state = t->out[0];
    return 0;
}
static int tr_add1(tr_t *t, fr_t e) {  /* :32-40 */
    t->n_out = 0; t->out_pos = 0;
    t->pending[t->n_pending++] = e;
    return t->n_pending == 16 ? tr_update(t) : 0;
}
int FH_X(tr_put1)(void *p, uint64_t v) { fr_t x = fr_zero(); x.l[0] = v; return tr_add1((tr_t *)p, fr_to_mont(x)); }   /* :92-93 */
int FH_X(tr_put4)(void *p, const uint64_t d[4]) { fr_t x; memcpy(&x, d, 32); return tr_add1((tr_t *)p, x); }              /* :94-97 raw digest */
static int tr_get253(tr_t *t, fr_t *o) {  /* :42-48 */
    if (t->out_pos >= t->n_out && tr_update(t)) return -1;
    *o = t->out[t->out_pos++];
    return 0;
}
int FH_X(tr_get_fields1)(void *p, uint64_t *o) {  /* :71-88 */
    tr_t *t = (tr_t *)p;
    const uint64_t GLP = 0xFFFFFFFF00000001ULL;
    for (;;) {
        if (t->out3_pos < t->n_out3) { *o = t->out3[t->out3_pos++]; return 0; }
        if (t->out_pos < t->n_out) {
            fr_t v = fr_from_mont(t->out[t->out_pos++]);
            for (int i = 0; i < 3; ++i) t->out3[i] = v.l[i] % GLP;   /* biguint_to_be (helper.rs:61-65) */
            t->out3_pos = 0; t->n_out3 = 3;
            continue;
        }
        if (tr_update(t)) return -1;
    }
}
int FH_X(tr_get_permutations)(void *p, uint32_t n, uint32_t nbits, uint64_t *out) {  /* :103-131 */
    tr_t *t = (tr_t *)p;
    const uint32_t total = n * nbits, nf = (total - 1) / 253 + 1;
    fr_t *f = (fr_t *)malloc(nf * sizeof(fr_t));
    for (uint32_t i = 0; i < nf; ++i) { fr_t v; if (tr_get253(t, &v)) { free(f); return -1; } f[i] = fr_from_mont(v); }
    uint32_t cf = 0, cb = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t a = 0;
        for (uint32_t j = 0; j < nbits; ++j) {
            if ((f[cf].l[cb / 64] >> (cb % 64)) & 1) a += (uint64_t)1 << j;
            if (++cb == 253) { cb = 0; ++cf; }
        }
        out[i] = a;
    }
    free(f);
    return 0;
}
