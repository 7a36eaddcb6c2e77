/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Field-generic body of the Groth16 prover's scalar-field work, included by
 * groth16_bn254.c / groth16_bls12_381.c after they define G16_X(name), G16_RMOD, G16_R2, G16_RINV, G16_S (2-adicity)
 * and G16_ROOT (a 2^S-th primitive root of unity, Montgomery form).
 *
 * The reference reaches this code only through a third-party dependency that is NOT under /root/reference:
 * bellman_ce 0.3.2 (Cargo.lock:668-670), call site groth16/src/groth16.rs:93 (create_random_proof).  What follows
 * restates bellman's published algorithm (domain.rs EvaluationDomain::{fft, ifft, coset_fft, icoset_fft,
 * divide_by_z_on_coset}, groth16/prover.rs create_proof):
 *   h(X) = (A(X) B(X) - C(X)) / (X^m - 1) from the per-constraint evaluations a_i, b_i, c_i:
 *   3 x ifft, 3 x coset_fft (coset generator g = Fr::multiplicative_generator() = 7), pointwise a*b - c,
 *   times (g^m - 1)^-1, icoset_fft, drop the top coefficient.
 * omega for a 2^k domain = ROOT^(2^(S-k)), ROOT = 7^((r-1)/2^S) (ff's derive macro).  The reference holds no proving
 * key, witness or quotient vector and its proofs are randomised, so no byte of a proof can be compared; the pins are
 * (tests/test_oracle_groth16.py, tests/test_oracle_pairing.py): the transform against the O(n^2) definition, h against
 * polynomial division, whole proofs against the verification equation evaluated in the exponent with a known trapdoor,
 * AND against the equation itself through oracle/pairing.py, a pairing verifier that accepts the reference's own
 * proof fixture (groth16/test-vectors/proof.json under verification_key.json, public input 33). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;
static const fr_t RMOD = {{G16_RMOD}};
static const fr_t R2 = {{G16_R2}};
static const fr_t ROOT = {{G16_ROOT}};
#define RINV G16_RINV

static int fr_geq(const fr_t *a, const fr_t *b) {
    for (int i = 3; i >= 0; --i) { if (a->l[i] > b->l[i]) return 1; if (a->l[i] < b->l[i]) return 0; }
    return 1;
}
static fr_t fr_add(fr_t a, fr_t b) {
    fr_t r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || fr_geq(&r, &RMOD)) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)r.l[i] - RMOD.l[i] - br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return r;
}
static fr_t fr_sub(fr_t a, fr_t b) {
    fr_t r; u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + RMOD.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static fr_t fr_mul(fr_t a, fr_t b) { /* Montgomery CIOS: a*b/2^256 mod r */
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
static fr_t fr_one(void) { fr_t one = {{1, 0, 0, 0}}; return fr_mul(one, R2); }
static fr_t fr_from_u64(uint64_t v) { fr_t a = {{v, 0, 0, 0}}; return fr_mul(a, R2); }
static fr_t fr_pow(fr_t a, const uint64_t e[4]) {
    fr_t r = fr_one();
    for (int i = 255; i >= 0; --i) { r = fr_mul(r, r); if ((e[i >> 6] >> (i & 63)) & 1) r = fr_mul(r, a); }
    return r;
}
static fr_t fr_inv(fr_t a) {
    uint64_t e[4] = {RMOD.l[0] - 2, RMOD.l[1], RMOD.l[2], RMOD.l[3]};
    return fr_pow(a, e);
}
static fr_t fr_pow_u64(fr_t a, uint64_t k) { uint64_t e[4] = {k, 0, 0, 0}; return fr_pow(a, e); }

void G16_X(fr_mul)(const uint64_t *a, const uint64_t *b, uint64_t *r) { fr_t x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); x = fr_mul(x, y); memcpy(r, &x, 32); }
void G16_X(fr_to_mont)(const uint64_t *a, uint64_t *r, uint64_t n) { for (uint64_t i = 0; i < n; ++i) { fr_t x; memcpy(&x, a + 4 * i, 32); x = fr_mul(x, R2); memcpy(r + 4 * i, &x, 32); } }
void G16_X(fr_from_mont)(const uint64_t *a, uint64_t *r, uint64_t n) { const fr_t one = {{1, 0, 0, 0}}; for (uint64_t i = 0; i < n; ++i) { fr_t x; memcpy(&x, a + 4 * i, 32); x = fr_mul(x, one); memcpy(r + 4 * i, &x, 32); } }

static fr_t domain_omega(unsigned log_n) { fr_t w = ROOT; for (unsigned i = log_n; i < G16_S; ++i) w = fr_mul(w, w); return w; }

/* domain.rs serial_fft: bit-reversal permutation, then log_n rounds of in-place butterflies */
static void fft_core(fr_t *a, unsigned log_n, fr_t omega) {
    const uint64_t n = 1ULL << log_n;
    for (uint64_t k = 0; k < n; ++k) {
        uint64_t rk = 0;
        for (unsigned b = 0; b < log_n; ++b) rk |= ((k >> b) & 1) << (log_n - 1 - b);
        if (k < rk) { fr_t t = a[k]; a[k] = a[rk]; a[rk] = t; }
    }
    uint64_t m = 1;
    for (unsigned s = 0; s < log_n; ++s) {
        const fr_t w_m = fr_pow_u64(omega, n / (2 * m));
        fr_t *tw = (fr_t *)malloc(m * sizeof(fr_t));
        tw[0] = fr_one();
        for (uint64_t j = 1; j < m; ++j) tw[j] = fr_mul(tw[j - 1], w_m);
#pragma omp parallel for schedule(static) if (n >= (1u << 15))
        for (uint64_t x = 0; x < n / 2; ++x) {
            const uint64_t j = x & (m - 1), k = (x - j) * 2;
            const fr_t t = fr_mul(a[k + j + m], tw[j]);
            a[k + j + m] = fr_sub(a[k + j], t);
            a[k + j] = fr_add(a[k + j], t);
        }
        free(tw);
        m *= 2;
    }
}
static void scale_all(fr_t *a, uint64_t n, fr_t s) {
#pragma omp parallel for schedule(static) if (n >= (1u << 15))
    for (uint64_t i = 0; i < n; ++i) a[i] = fr_mul(a[i], s);
}
static void distribute_powers(fr_t *a, uint64_t n, fr_t g) {   /* a[i] *= g^i */
    fr_t *p = (fr_t *)malloc(n * sizeof(fr_t));
    p[0] = fr_one();
    for (uint64_t i = 1; i < n; ++i) p[i] = fr_mul(p[i - 1], g);
#pragma omp parallel for schedule(static) if (n >= (1u << 15))
    for (uint64_t i = 0; i < n; ++i) a[i] = fr_mul(a[i], p[i]);
    free(p);
}
static void do_fft(fr_t *a, unsigned log_n) { fft_core(a, log_n, domain_omega(log_n)); }
static void do_ifft(fr_t *a, unsigned log_n) {
    fft_core(a, log_n, fr_inv(domain_omega(log_n)));
    scale_all(a, 1ULL << log_n, fr_inv(fr_from_u64(1ULL << log_n)));
}

/* EvaluationDomain::{fft, ifft, coset_fft, icoset_fft}; data n x 4 u64 Montgomery, natural order in and out */
void G16_X(fr_ntt)(uint64_t *data, unsigned log_n, int inverse, int coset) {
    fr_t *a = (fr_t *)data;
    const uint64_t n = 1ULL << log_n;
    const fr_t g = fr_from_u64(7);
    if (!inverse) { if (coset) distribute_powers(a, n, g); do_fft(a, log_n); }
    else { do_ifft(a, log_n); if (coset) distribute_powers(a, n, fr_inv(g)); }
}

/* prover.rs create_proof, the `h` block: a <- coefficients of (A B - C)/Z (all n written; the caller drops the last) */
void G16_X(quotient)(uint64_t *a_, uint64_t *b_, uint64_t *c_, unsigned log_n) {
    fr_t *a = (fr_t *)a_, *b = (fr_t *)b_, *c = (fr_t *)c_;
    const uint64_t n = 1ULL << log_n;
    G16_X(fr_ntt)(a_, log_n, 1, 0); G16_X(fr_ntt)(a_, log_n, 0, 1);
    G16_X(fr_ntt)(b_, log_n, 1, 0); G16_X(fr_ntt)(b_, log_n, 0, 1);
    G16_X(fr_ntt)(c_, log_n, 1, 0); G16_X(fr_ntt)(c_, log_n, 0, 1);
    const fr_t zinv = fr_inv(fr_sub(fr_pow_u64(fr_from_u64(7), n), fr_one()));   /* divide_by_z_on_coset */
#pragma omp parallel for schedule(static) if (n >= (1u << 15))
    for (uint64_t i = 0; i < n; ++i) a[i] = fr_mul(fr_sub(fr_mul(a[i], b[i]), c[i]), zinv);
    G16_X(fr_ntt)(a_, log_n, 1, 1);
}

/* ProvingAssignment::enforce's eval(): out[i] = sum_k coeff[k] * w[col[k]] over row i of a CSR matrix (Montgomery) */
void G16_X(r1cs_eval)(const uint64_t *row_ptr, const uint32_t *cols, const uint64_t *coeffs, const uint64_t *w, uint64_t n_rows, uint64_t *out) {
#pragma omp parallel for schedule(static) if (n_rows >= (1u << 15))
    for (uint64_t i = 0; i < n_rows; ++i) {
        fr_t acc; memset(&acc, 0, sizeof acc);
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            fr_t cf, x; memcpy(&cf, coeffs + 4 * k, 32); memcpy(&x, w + 4 * (uint64_t)cols[k], 32);
            acc = fr_add(acc, fr_mul(cf, x));
        }
        memcpy(out + 4 * i, &acc, 32);
    }
}
