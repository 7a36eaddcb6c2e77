/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Curve-generic part of the G1 MSM restatement: included by ec.c
 * (BN254) and ec_bls12_381.c (BLS12-381) after they define
 *   EC_NL            64-bit limbs of Fq (Montgomery R = 2^(64*EC_NL))
 *   EC_Q, EC_R1, EC_R2, EC_QINV   modulus, R mod q, R^2 mod q, -q^-1 mod 2^64
 *   EC_B             the curve constant of y^2 = x^3 + b (small integer)
 *   EC_GX, EC_GY     generator, canonical (non-Montgomery) limbs
 *   EC_X(name)       exported symbol prefix
 * Scalars are 4 x u64 canonical little-endian (FrRepr) on both curves.  See ec.c for provenance. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[EC_NL]; } fq_t;
typedef struct { fq_t x, y; int inf; } aff_t;
typedef struct { fq_t x, y, z; } jac_t;   /* z == 0 <=> infinity */
#define FQB (8 * EC_NL)                    /* bytes per field element */
#define PTW (2 * EC_NL)                    /* words per affine point */

static const fq_t Q = {EC_Q};
static const fq_t R1 = {EC_R1};
static const fq_t R2 = {EC_R2};
#define QINV EC_QINV

static int fq_geq(const fq_t *a, const fq_t *b) {
    for (int i = EC_NL - 1; i >= 0; --i) { if (a->l[i] > b->l[i]) return 1; if (a->l[i] < b->l[i]) return 0; }
    return 1;
}
static void fq_sub_nored(fq_t *r, const fq_t *a, const fq_t *b) {
    u128 br = 0;
    for (int i = 0; i < EC_NL; ++i) { u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static fq_t fq_add(fq_t a, fq_t b) {
    fq_t r; u128 c = 0;
    for (int i = 0; i < EC_NL; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || fq_geq(&r, &Q)) fq_sub_nored(&r, &r, &Q);
    return r;
}
static fq_t fq_sub(fq_t a, fq_t b) {
    fq_t r;
    if (fq_geq(&a, &b)) fq_sub_nored(&r, &a, &b);
    else { fq_t t; fq_sub_nored(&t, &Q, &b); u128 c = 0; for (int i = 0; i < EC_NL; ++i) { c += (u128)a.l[i] + t.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static fq_t fq_mul(fq_t a, fq_t b) { /* Montgomery CIOS */
    uint64_t t[EC_NL + 2]; memset(t, 0, sizeof t);
    for (int i = 0; i < EC_NL; ++i) {
        u128 c = 0;
        for (int j = 0; j < EC_NL; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[EC_NL]; t[EC_NL] = (uint64_t)c; t[EC_NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * QINV;
        c = ((u128)m * Q.l[0] + t[0]) >> 64;
        for (int j = 1; j < EC_NL; ++j) { c += (u128)m * Q.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[EC_NL]; t[EC_NL - 1] = (uint64_t)c; t[EC_NL] = t[EC_NL + 1] + (uint64_t)(c >> 64);
    }
    fq_t r; memcpy(&r, t, sizeof r);
    if (t[EC_NL] || fq_geq(&r, &Q)) fq_sub_nored(&r, &r, &Q);
    return r;
}
static fq_t fq_sqr(fq_t a) { return fq_mul(a, a); }
static int fq_is_zero(fq_t a) { uint64_t o = 0; for (int i = 0; i < EC_NL; ++i) o |= a.l[i]; return !o; }
static int fq_eq(fq_t a, fq_t b) { return !memcmp(&a, &b, sizeof a); }
static fq_t fq_inv(fq_t a) { /* a^(q-2) */
    fq_t e = Q, r = R1; e.l[0] -= 2;
    for (int i = 64 * EC_NL - 1; i >= 0; --i) { r = fq_sqr(r); if ((e.l[i / 64] >> (i % 64)) & 1) r = fq_mul(r, a); }
    return r;
}
static fq_t fq_from_u64(uint64_t v) { fq_t a; memset(&a, 0, sizeof a); a.l[0] = v; return fq_mul(a, R2); }

static jac_t jac_inf(void) { jac_t p; memset(&p, 0, sizeof p); return p; }
static jac_t jac_dbl(jac_t p) { /* dbl-2009-l, a = 0 */
    if (fq_is_zero(p.z)) return p;
    fq_t A = fq_sqr(p.x), B = fq_sqr(p.y), C = fq_sqr(B);
    fq_t D = fq_sub(fq_sub(fq_sqr(fq_add(p.x, B)), A), C); D = fq_add(D, D);
    fq_t E = fq_add(fq_add(A, A), A), F = fq_sqr(E);
    jac_t r;
    r.x = fq_sub(F, fq_add(D, D));
    fq_t C8 = fq_add(C, C); C8 = fq_add(C8, C8); C8 = fq_add(C8, C8);
    r.y = fq_sub(fq_mul(E, fq_sub(D, r.x)), C8);
    r.z = fq_mul(p.y, p.z); r.z = fq_add(r.z, r.z);
    return r;
}
static jac_t jac_add_aff(jac_t p, const aff_t *q) { /* madd-2007-bl */
    if (q->inf) return p;
    if (fq_is_zero(p.z)) { jac_t r; r.x = q->x; r.y = q->y; r.z = R1; return r; }
    fq_t Z1Z1 = fq_sqr(p.z), U2 = fq_mul(q->x, Z1Z1), S2 = fq_mul(fq_mul(q->y, p.z), Z1Z1);
    if (fq_eq(U2, p.x)) { if (fq_eq(S2, p.y)) return jac_dbl(p); return jac_inf(); }
    fq_t H = fq_sub(U2, p.x), HH = fq_sqr(H), I = fq_add(HH, HH); I = fq_add(I, I);
    fq_t J = fq_mul(H, I), r_ = fq_sub(S2, p.y); r_ = fq_add(r_, r_);
    fq_t V = fq_mul(p.x, I);
    jac_t r;
    r.x = fq_sub(fq_sub(fq_sqr(r_), J), fq_add(V, V));
    fq_t YJ = fq_mul(p.y, J);
    r.y = fq_sub(fq_mul(r_, fq_sub(V, r.x)), fq_add(YJ, YJ));
    r.z = fq_sub(fq_sub(fq_sqr(fq_add(p.z, H)), Z1Z1), HH);
    return r;
}
static jac_t jac_add(jac_t p, jac_t q) { /* add-2007-bl */
    if (fq_is_zero(p.z)) return q;
    if (fq_is_zero(q.z)) return p;
    fq_t Z1Z1 = fq_sqr(p.z), Z2Z2 = fq_sqr(q.z);
    fq_t U1 = fq_mul(p.x, Z2Z2), U2 = fq_mul(q.x, Z1Z1);
    fq_t S1 = fq_mul(fq_mul(p.y, q.z), Z2Z2), S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1);
    if (fq_eq(U1, U2)) { if (fq_eq(S1, S2)) return jac_dbl(p); return jac_inf(); }
    fq_t H = fq_sub(U2, U1), I = fq_sqr(fq_add(H, H)), J = fq_mul(H, I);
    fq_t r_ = fq_sub(S2, S1); r_ = fq_add(r_, r_);
    fq_t V = fq_mul(U1, I);
    jac_t r;
    r.x = fq_sub(fq_sub(fq_sqr(r_), J), fq_add(V, V));
    fq_t SJ = fq_mul(S1, J);
    r.y = fq_sub(fq_mul(r_, fq_sub(V, r.x)), fq_add(SJ, SJ));
    r.z = fq_mul(fq_sub(fq_sub(fq_sqr(fq_add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
static aff_t jac_to_aff(jac_t p) {
    aff_t a; memset(&a, 0, sizeof a);
    if (fq_is_zero(p.z)) { a.inf = 1; return a; }
    fq_t zi = fq_inv(p.z), zi2 = fq_sqr(zi);
    a.x = fq_mul(p.x, zi2); a.y = fq_mul(p.y, fq_mul(zi2, zi));
    return a;
}

static fq_t fq_load(const uint64_t *p) { fq_t x; memcpy(&x, p, FQB); return x; }

/* ---- exported ------------------------------------------------------------------------------- */
/* generator in Montgomery form */
void EC_X(generator)(uint64_t *out) {
    const fq_t gx = {EC_GX}, gy = {EC_GY};
    fq_t x = fq_mul(gx, R2), y = fq_mul(gy, R2);
    memcpy(out, &x, FQB); memcpy(out + EC_NL, &y, FQB);
}
int EC_X(on_curve)(const uint64_t *p) {
    fq_t x = fq_load(p), y = fq_load(p + EC_NL);
    return fq_eq(fq_sqr(y), fq_add(fq_mul(fq_sqr(x), x), fq_from_u64(EC_B)));
}
void EC_X(fq_mul)(const uint64_t *a, const uint64_t *b, uint64_t *r) { fq_t x = fq_mul(fq_load(a), fq_load(b)); memcpy(r, &x, FQB); }
void EC_X(fq_from_mont)(const uint64_t *a, uint64_t *r) { fq_t one; memset(&one, 0, sizeof one); one.l[0] = 1; fq_t x = fq_mul(fq_load(a), one); memcpy(r, &x, FQB); }
void EC_X(fq_to_mont)(const uint64_t *a, uint64_t *r) { fq_t x = fq_mul(fq_load(a), R2); memcpy(r, &x, FQB); }

/* [k]P, k = 256-bit little-endian; out: x || y (Montgomery), returns 1 if infinity */
int EC_X(scalar_mul)(const uint64_t *p, const uint64_t k[4], uint64_t *out) {
    aff_t a; a.x = fq_load(p); a.y = fq_load(p + EC_NL); a.inf = 0;
    jac_t acc = jac_inf();
    for (int i = 255; i >= 0; --i) { acc = jac_dbl(acc); if ((k[i / 64] >> (i % 64)) & 1) acc = jac_add_aff(acc, &a); }
    aff_t r = jac_to_aff(acc);
    memcpy(out, &r.x, FQB); memcpy(out + EC_NL, &r.y, FQB);
    return r.inf;
}

/* bases[i] = [a + i*b]G for i < n (a, b 64-bit): one addition per base + batched affine conversion */
void EC_X(make_bases)(uint64_t n, uint64_t a, uint64_t b, uint64_t *out /* n * PTW */) {
    uint64_t g[PTW], tmp[PTW], ka[4] = {a, 0, 0, 0}, kb[4] = {b, 0, 0, 0};
    EC_X(generator)(g);
    aff_t step; EC_X(scalar_mul)(g, kb, tmp); step.x = fq_load(tmp); step.y = fq_load(tmp + EC_NL); step.inf = b == 0;   /* b = 0: [0]G is the point at infinity, every base is [a]G (round 6: the fuzz drew b = 0 and got garbage bases) */
    aff_t first; EC_X(scalar_mul)(g, ka, tmp); first.x = fq_load(tmp); first.y = fq_load(tmp + EC_NL); first.inf = 0;
    jac_t cur; cur.x = first.x; cur.y = first.y; cur.z = R1;
    jac_t *pts = (jac_t *)malloc(n * sizeof(jac_t));
    fq_t *pre = (fq_t *)malloc(n * sizeof(fq_t));
    for (uint64_t i = 0; i < n; ++i) { pts[i] = cur; cur = jac_add_aff(cur, &step); }
    fq_t acc = R1;                                           /* Montgomery batch inversion of all z */
    for (uint64_t i = 0; i < n; ++i) { pre[i] = acc; acc = fq_mul(acc, pts[i].z); }
    fq_t inv = fq_inv(acc);
    for (uint64_t i = n; i-- > 0;) {
        fq_t zi = fq_mul(inv, pre[i]); inv = fq_mul(inv, pts[i].z);
        fq_t zi2 = fq_sqr(zi), x = fq_mul(pts[i].x, zi2), y = fq_mul(pts[i].y, fq_mul(zi2, zi));
        memcpy(out + PTW * i, &x, FQB); memcpy(out + PTW * i + EC_NL, &y, FQB);
    }
    free(pts); free(pre);
}

/* Pippenger bucket method, window c bits.  scalars: n*4 words canonical LE.  returns inf flag */
int EC_X(msm)(const uint64_t *bases, const uint64_t *scalars, uint64_t n, unsigned c, uint64_t *out) {
    unsigned nw = (256 + c - 1) / c;
    size_t nb = ((size_t)1 << c) - 1;
    jac_t *wres = (jac_t *)malloc(nw * sizeof(jac_t));
    #pragma omp parallel for schedule(dynamic, 1)
    for (unsigned w = 0; w < nw; ++w) {
        jac_t *bk = (jac_t *)calloc(nb, sizeof(jac_t));
        for (uint64_t i = 0; i < n; ++i) {
            unsigned bit = w * c; uint64_t limb = bit / 64, off = bit % 64;
            if (limb > 3) continue;
            uint64_t v = scalars[4 * i + limb] >> off;
            if (off + c > 64 && limb < 3) v |= scalars[4 * i + limb + 1] << (64 - off);
            v &= ((uint64_t)1 << c) - 1;
            if (!v) continue;
            aff_t a; a.x = fq_load(bases + PTW * i); a.y = fq_load(bases + PTW * i + EC_NL); a.inf = 0;
            bk[v - 1] = jac_add_aff(bk[v - 1], &a);
        }
        jac_t run = jac_inf(), sum = jac_inf();
        for (size_t k = nb; k-- > 0;) { run = jac_add(run, bk[k]); sum = jac_add(sum, run); }
        wres[w] = sum;
        free(bk);
    }
    jac_t acc = jac_inf();
    for (unsigned w = nw; w-- > 0;) { for (unsigned k = 0; k < c; ++k) acc = jac_dbl(acc); acc = jac_add(acc, wres[w]); }
    free(wres);
    aff_t r = jac_to_aff(acc);
    memcpy(out, &r.x, FQB); memcpy(out + EC_NL, &r.y, FQB);
    return r.inf;
}

/* cpu_baseline of bench.py's MSM lines: the same bucket method with the (window, chunk-of-points) pairs spread over every
 * host thread (bellman's multiexp splits the work over its worker pool the same way); equals EC_X(msm) point for point. */
int EC_X(msm_par)(const uint64_t *bases, const uint64_t *scalars, uint64_t n, unsigned c, unsigned chunks, uint64_t *out) {
    unsigned nw = (256 + c - 1) / c;
    size_t nb = ((size_t)1 << c) - 1;
    if (chunks < 1) chunks = 1;
    jac_t *wres = (jac_t *)malloc((size_t)nw * chunks * sizeof(jac_t));
    #pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (unsigned w = 0; w < nw; ++w)
        for (unsigned ch = 0; ch < chunks; ++ch) {
            const uint64_t lo = n * ch / chunks, hi = n * (ch + 1) / chunks;
            jac_t *bk = (jac_t *)calloc(nb, sizeof(jac_t));
            for (uint64_t i = lo; i < hi; ++i) {
                unsigned bit = w * c; uint64_t limb = bit / 64, off = bit % 64;
                if (limb > 3) continue;
                uint64_t v = scalars[4 * i + limb] >> off;
                if (off + c > 64 && limb < 3) v |= scalars[4 * i + limb + 1] << (64 - off);
                v &= ((uint64_t)1 << c) - 1;
                if (!v) continue;
                aff_t a; a.x = fq_load(bases + PTW * i); a.y = fq_load(bases + PTW * i + EC_NL); a.inf = 0;
                bk[v - 1] = jac_add_aff(bk[v - 1], &a);
            }
            jac_t run = jac_inf(), sum = jac_inf();
            for (size_t k = nb; k-- > 0;) { run = jac_add(run, bk[k]); sum = jac_add(sum, run); }
            wres[(size_t)w * chunks + ch] = sum;
            free(bk);
        }
    jac_t acc = jac_inf();
    for (unsigned w = nw; w-- > 0;) {
        for (unsigned k = 0; k < c; ++k) acc = jac_dbl(acc);
        for (unsigned ch = 0; ch < chunks; ++ch) acc = jac_add(acc, wres[(size_t)w * chunks + ch]);
    }
    free(wres);
    aff_t r = jac_to_aff(acc);
    memcpy(out, &r.x, FQB); memcpy(out + EC_NL, &r.y, FQB);
    return r.inf;
}

/* ---- G2: the sextic twist y^2 = x^3 + b' over Fq2 = Fq[u]/(u^2 + 1) ---------------------------------------
 * Needs EC_B2C0, EC_B2C1 (b' = c0 + c1 u, canonical limbs) and EC_G2X0/X1/Y0/Y1 (generator, canonical limbs).
 * Points travel as pairing_ce keeps G2Affine: x.c0 || x.c1 || y.c0 || y.c1, each Fq in Montgomery form. */
typedef struct { fq_t c0, c1; } fq2_t;
typedef struct { fq2_t x, y; int inf; } aff2_t;
typedef struct { fq2_t x, y, z; } jac2_t;
#define PT2W (4 * EC_NL)
static fq_t fq_neg(fq_t a) { fq_t z; memset(&z, 0, sizeof z); return fq_sub(z, a); }
static fq2_t f2_add(fq2_t a, fq2_t b) { fq2_t r = {fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; return r; }
static fq2_t f2_sub(fq2_t a, fq2_t b) { fq2_t r = {fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; return r; }
static fq2_t f2_mul(fq2_t a, fq2_t b) {
    fq2_t r = {fq_sub(fq_mul(a.c0, b.c0), fq_mul(a.c1, b.c1)), fq_add(fq_mul(a.c0, b.c1), fq_mul(a.c1, b.c0))};
    return r;
}
static fq2_t f2_sqr(fq2_t a) { return f2_mul(a, a); }
static int f2_is_zero(fq2_t a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
static int f2_eq(fq2_t a, fq2_t b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
static fq2_t f2_inv(fq2_t a) {
    fq_t n = fq_inv(fq_add(fq_sqr(a.c0), fq_sqr(a.c1)));
    fq2_t r = {fq_mul(a.c0, n), fq_mul(fq_neg(a.c1), n)};
    return r;
}
static fq2_t f2_load(const uint64_t *p) { fq2_t x; x.c0 = fq_load(p); x.c1 = fq_load(p + EC_NL); return x; }
static void f2_store(uint64_t *p, fq2_t x) { memcpy(p, &x.c0, FQB); memcpy(p + EC_NL, &x.c1, FQB); }
static fq2_t f2_one(void) { fq2_t r; r.c0 = R1; memset(&r.c1, 0, sizeof r.c1); return r; }

static jac2_t jac2_inf(void) { jac2_t p; memset(&p, 0, sizeof p); return p; }
static jac2_t jac2_dbl(jac2_t p) { /* dbl-2009-l, a = 0 */
    if (f2_is_zero(p.z)) return p;
    fq2_t A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
    fq2_t D = f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), C); D = f2_add(D, D);
    fq2_t E = f2_add(f2_add(A, A), A), F = f2_sqr(E);
    jac2_t r;
    r.x = f2_sub(F, f2_add(D, D));
    fq2_t C8 = f2_add(C, C); C8 = f2_add(C8, C8); C8 = f2_add(C8, C8);
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), C8);
    r.z = f2_mul(p.y, p.z); r.z = f2_add(r.z, r.z);
    return r;
}
static jac2_t jac2_add_aff(jac2_t p, const aff2_t *q) { /* madd-2007-bl */
    if (q->inf) return p;
    if (f2_is_zero(p.z)) { jac2_t r; r.x = q->x; r.y = q->y; r.z = f2_one(); return r; }
    fq2_t Z1Z1 = f2_sqr(p.z), U2 = f2_mul(q->x, Z1Z1), S2 = f2_mul(f2_mul(q->y, p.z), Z1Z1);
    if (f2_eq(U2, p.x)) { if (f2_eq(S2, p.y)) return jac2_dbl(p); return jac2_inf(); }
    fq2_t H = f2_sub(U2, p.x), HH = f2_sqr(H), I = f2_add(HH, HH); I = f2_add(I, I);
    fq2_t J = f2_mul(H, I), r_ = f2_sub(S2, p.y); r_ = f2_add(r_, r_);
    fq2_t V = f2_mul(p.x, I);
    jac2_t r;
    r.x = f2_sub(f2_sub(f2_sqr(r_), J), f2_add(V, V));
    fq2_t YJ = f2_mul(p.y, J);
    r.y = f2_sub(f2_mul(r_, f2_sub(V, r.x)), f2_add(YJ, YJ));
    r.z = f2_sub(f2_sub(f2_sqr(f2_add(p.z, H)), Z1Z1), HH);
    return r;
}
static jac2_t jac2_add(jac2_t p, jac2_t q) { /* add-2007-bl */
    if (f2_is_zero(p.z)) return q;
    if (f2_is_zero(q.z)) return p;
    fq2_t Z1Z1 = f2_sqr(p.z), Z2Z2 = f2_sqr(q.z);
    fq2_t U1 = f2_mul(p.x, Z2Z2), U2 = f2_mul(q.x, Z1Z1);
    fq2_t S1 = f2_mul(f2_mul(p.y, q.z), Z2Z2), S2 = f2_mul(f2_mul(q.y, p.z), Z1Z1);
    if (f2_eq(U1, U2)) { if (f2_eq(S1, S2)) return jac2_dbl(p); return jac2_inf(); }
    fq2_t H = f2_sub(U2, U1), I = f2_sqr(f2_add(H, H)), J = f2_mul(H, I);
    fq2_t r_ = f2_sub(S2, S1); r_ = f2_add(r_, r_);
    fq2_t V = f2_mul(U1, I);
    jac2_t r;
    r.x = f2_sub(f2_sub(f2_sqr(r_), J), f2_add(V, V));
    fq2_t SJ = f2_mul(S1, J);
    r.y = f2_sub(f2_mul(r_, f2_sub(V, r.x)), f2_add(SJ, SJ));
    r.z = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
static aff2_t jac2_to_aff(jac2_t p) {
    aff2_t a; memset(&a, 0, sizeof a);
    if (f2_is_zero(p.z)) { a.inf = 1; return a; }
    fq2_t zi = f2_inv(p.z), zi2 = f2_sqr(zi);
    a.x = f2_mul(p.x, zi2); a.y = f2_mul(p.y, f2_mul(zi2, zi));
    return a;
}
static aff2_t aff2_load(const uint64_t *p) { aff2_t a; a.x = f2_load(p); a.y = f2_load(p + 2 * EC_NL); a.inf = 0; return a; }

void EC_X(g2_generator)(uint64_t *out) {
    const fq_t x0 = {EC_G2X0}, x1 = {EC_G2X1}, y0 = {EC_G2Y0}, y1 = {EC_G2Y1};
    fq2_t x = {fq_mul(x0, R2), fq_mul(x1, R2)}, y = {fq_mul(y0, R2), fq_mul(y1, R2)};
    f2_store(out, x); f2_store(out + 2 * EC_NL, y);
}
int EC_X(g2_on_curve)(const uint64_t *p) {
    const fq_t b0 = {EC_B2C0}, b1 = {EC_B2C1};
    fq2_t b = {fq_mul(b0, R2), fq_mul(b1, R2)};
    fq2_t x = f2_load(p), y = f2_load(p + 2 * EC_NL);
    return f2_eq(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), b));
}
int EC_X(g2_scalar_mul)(const uint64_t *p, const uint64_t k[4], uint64_t *out) {
    aff2_t a = aff2_load(p);
    jac2_t acc = jac2_inf();
    for (int i = 255; i >= 0; --i) { acc = jac2_dbl(acc); if ((k[i / 64] >> (i % 64)) & 1) acc = jac2_add_aff(acc, &a); }
    aff2_t r = jac2_to_aff(acc);
    f2_store(out, r.x); f2_store(out + 2 * EC_NL, r.y);
    return r.inf;
}
void EC_X(g2_make_bases)(uint64_t n, uint64_t a, uint64_t b, uint64_t *out /* n * PT2W */) {
    uint64_t g[PT2W], tmp[PT2W], ka[4] = {a, 0, 0, 0}, kb[4] = {b, 0, 0, 0};
    EC_X(g2_generator)(g);
    EC_X(g2_scalar_mul)(g, kb, tmp); aff2_t step = aff2_load(tmp); if (b == 0) step.inf = 1;
    EC_X(g2_scalar_mul)(g, ka, tmp); aff2_t first = aff2_load(tmp);
    jac2_t cur; cur.x = first.x; cur.y = first.y; cur.z = f2_one();
    for (uint64_t i = 0; i < n; ++i) {              /* sizes used in tests are small: one inversion per point */
        aff2_t p = jac2_to_aff(cur);
        f2_store(out + PT2W * i, p.x); f2_store(out + PT2W * i + 2 * EC_NL, p.y);
        cur = jac2_add_aff(cur, &step);
    }
}
int EC_X(g2_msm)(const uint64_t *bases, const uint64_t *scalars, uint64_t n, unsigned c, uint64_t *out) {
    unsigned nw = (256 + c - 1) / c;
    size_t nb = ((size_t)1 << c) - 1;
    jac2_t *wres = (jac2_t *)malloc(nw * sizeof(jac2_t));
    #pragma omp parallel for schedule(dynamic, 1)
    for (unsigned w = 0; w < nw; ++w) {
        jac2_t *bk = (jac2_t *)calloc(nb, sizeof(jac2_t));
        for (uint64_t i = 0; i < n; ++i) {
            unsigned bit = w * c; uint64_t limb = bit / 64, off = bit % 64;
            if (limb > 3) continue;
            uint64_t v = scalars[4 * i + limb] >> off;
            if (off + c > 64 && limb < 3) v |= scalars[4 * i + limb + 1] << (64 - off);
            v &= ((uint64_t)1 << c) - 1;
            if (!v) continue;
            aff2_t a = aff2_load(bases + PT2W * i);
            bk[v - 1] = jac2_add_aff(bk[v - 1], &a);
        }
        jac2_t run = jac2_inf(), sum = jac2_inf();
        for (size_t k = nb; k-- > 0;) { run = jac2_add(run, bk[k]); sum = jac2_add(sum, run); }
        wres[w] = sum;
        free(bk);
    }
    jac2_t acc = jac2_inf();
    for (unsigned w = nw; w-- > 0;) { for (unsigned k = 0; k < c; ++k) acc = jac2_dbl(acc); acc = jac2_add(acc, wres[w]); }
    free(wres);
    aff2_t r = jac2_to_aff(acc);
    f2_store(out, r.x); f2_store(out + 2 * EC_NL, r.y);
    return r.inf;
}
