/* ORACLE -- TEST INFRASTRUCTURE ONLY (see gl.h).  CPU restatement of the per-stage glue of the
 * starky prover: FRI folding, evaluation tables, x/(x-xi), quotient split, grand products.
 * F3G values are 3 consecutive u64 (never the reference's 32-byte tagged struct).              */
#include "gl.h"
#include <stdlib.h>
#include <string.h>

uint32_t orc_bitrev(uint32_t x, unsigned bits);

static f3_t ld3(const uint64_t *p) { f3_t r; memcpy(r.v, p, 24); return r; }
static void st3(uint64_t *p, f3_t v) { memcpy(p, v.v, 24); }

/* fft.rs:39-83 over F3G with base-field roots: natural in/out; inverse includes 1/n + reversal */
static void f3_ntt(f3_t *buf, unsigned bits, int inverse) {
    size_t n = (size_t)1 << bits;
    for (size_t i = 0; i < n; ++i) {
        size_t r = orc_bitrev((uint32_t)i, bits);
        if (r > i) { f3_t t = buf[i]; buf[i] = buf[r]; buf[r] = t; }
    }
    for (unsigned s = 1; s <= bits; ++s) {
        size_t m = (size_t)1 << s, md2 = m >> 1;
        uint64_t winc = gl_root(s);
        for (size_t k = 0; k < n; k += m) {
            uint64_t w = 1;
            for (size_t j = 0; j < md2; ++j) {
                f3_t t = f3_muls(buf[k + j + md2], w), u = buf[k + j];
                buf[k + j] = f3_add(u, t);
                buf[k + j + md2] = f3_sub(u, t);
                w = gl_mul(w, winc);
            }
        }
    }
    if (inverse) {
        uint64_t ninv = gl_inv(gl_red((uint64_t)n));
        f3_t *res = (f3_t *)malloc(n * sizeof(f3_t));
        res[0] = f3_muls(buf[0], ninv);
        for (size_t i = 1; i < n; ++i) res[i] = f3_muls(buf[n - i], ninv);
        memcpy(buf, res, n * sizeof(f3_t));
        free(res);
    }
}
void orc_f3_ntt(uint64_t *buf /* [n][3] in place */, unsigned bits, int inverse) { f3_ntt((f3_t *)buf, bits, inverse); }

/* fri.rs:101-126 -- one folding step.  pol: [2^pol_bits][3]; out: [2^step_bits][3].
 * shift_inv = (49^-1)^(2^(nBitsExt - pol_bits)) is passed in (fri.rs:96,147-150).               */
void orc_fri_fold(const uint64_t *pol, unsigned pol_bits, unsigned step_bits, const uint64_t special_x[3],
                  uint64_t shift_inv, uint64_t *out) {
    unsigned rbits = pol_bits - step_bits;
    size_t pol2_n = (size_t)1 << step_bits, n_x = (size_t)1 << rbits;
    f3_t sx = ld3(special_x);
    uint64_t sinv = shift_inv, wi = gl_inv(gl_root(pol_bits));
    f3_t *ppar = (f3_t *)malloc(n_x * sizeof(f3_t));
    for (size_t g = 0; g < pol2_n; ++g) {
        if (rbits == 0) { st3(out + 3 * g, ld3(pol + 3 * g)); continue; }   /* fri.rs:113-114 */
        for (size_t i = 0; i < n_x; ++i) ppar[i] = ld3(pol + 3 * (i * pol2_n + g));
        f3_ntt(ppar, rbits, 1);                                              /* ifft :120 */
        uint64_t r = 1;                                                      /* pol_mul_axi(ppar_c, 1, sinv) */
        for (size_t i = 0; i < n_x; ++i) { ppar[i] = f3_muls(ppar[i], r); r = gl_mul(r, sinv); }
        f3_t res = ppar[n_x - 1];                                            /* eval_pol polutils.rs:13-23 */
        for (size_t i = n_x - 1; i-- > 0;) res = f3_add(f3_mul(res, sx), ppar[i]);
        st3(out + 3 * g, res);
        sinv = gl_mul(sinv, wi);
    }
    free(ppar);
}

/* fri.rs:299-317 get_transposed_buffer: [n] F3G -> [w = 2^tbits][h = n/w][3] */
void orc_fri_transpose(const uint64_t *pol, uint64_t n, unsigned tbits, uint64_t *out) {
    uint64_t w = (uint64_t)1 << tbits, h = n / w;
    for (uint64_t i = 0; i < w; ++i)
        for (uint64_t j = 0; j < h; ++j) memcpy(out + (i * h + j) * 3, pol + (j * w + i) * 3, 24);
}

/* polutils.rs:35-53 batch_inverse over F3G (sequential Montgomery trick) */
void orc_f3_batch_inverse(const uint64_t *in, uint64_t n, uint64_t *out) {
    if (!n) return;
    f3_t *tmp = (f3_t *)malloc(n * sizeof(f3_t));
    tmp[0] = ld3(in);
    for (uint64_t i = 1; i < n; ++i) tmp[i] = f3_mul(ld3(in + 3 * i), tmp[i - 1]);
    f3_t z = f3_inv(tmp[n - 1]);
    for (uint64_t i = n - 1; i >= 1; --i) { st3(out + 3 * i, f3_mul(z, tmp[i - 1])); z = f3_mul(z, ld3(in + 3 * i)); }
    st3(out, z);
    free(tmp);
}

/* stark_gen.rs:481-522: x/(x - xi) for x = 49 * w_ext^k, k < 2^nbits_ext -> [Next][3] */
void orc_xdivxsub(const uint64_t xi[3], unsigned nbits_ext, uint64_t *out) {
    uint64_t n = (uint64_t)1 << nbits_ext, w = gl_root(nbits_ext), x = GL_SHIFT;
    uint64_t *den = (uint64_t *)malloc(n * 24), *inv = (uint64_t *)malloc(n * 24), *xs = (uint64_t *)malloc(n * 8);
    for (uint64_t k = 0; k < n; ++k) {
        xs[k] = x;
        den[3 * k] = gl_sub(x, xi[0]); den[3 * k + 1] = gl_neg(xi[1]); den[3 * k + 2] = gl_neg(xi[2]);
        x = gl_mul(x, w);
    }
    orc_f3_batch_inverse(den, n, inv);
    for (uint64_t k = 0; k < n; ++k) st3(out + 3 * k, f3_muls(ld3(inv + 3 * k), xs[k]));
    free(den); free(inv); free(xs);
}

/* stark_gen.rs:575-592 build_Zh_Inv: ZHInv[j] = 1 / (49^(2^nbits) * w_ext^j - 1), j < 2^ext */
void orc_zh_inv(unsigned nbits, unsigned extend_bits, uint64_t *out) {
    uint64_t sn = GL_SHIFT, w = 1, we = gl_root(extend_bits);
    for (unsigned i = 0; i < nbits; ++i) sn = gl_mul(sn, sn);
    for (uint64_t j = 0; j < ((uint64_t)1 << extend_bits); ++j) { out[j] = gl_inv(gl_sub(gl_mul(sn, w), 1)); w = gl_mul(w, we); }
}

/* stark_gen.rs:416-430: LEv[i] = (xi/49)^i (prime: (xi*w/49)^i), then FFT::ifft -> [N][3] */
void orc_lev(const uint64_t xi[3], unsigned nbits, int prime, uint64_t *out) {
    uint64_t n = (uint64_t)1 << nbits;
    f3_t x = f3_muls(ld3(xi), gl_inv(GL_SHIFT));
    if (prime) x = f3_muls(x, gl_root(nbits));
    f3_t *l = (f3_t *)out, cur = f3_from(1);
    for (uint64_t i = 0; i < n; ++i) { l[i] = cur; cur = f3_mul(cur, x); }
    f3_ntt(l, nbits, 1);
}

/* stark_gen.rs:450-466: sum_k col[(k << ext) * width + offset] * L[k]; col cell has dim 1 or 3 */
void orc_eval_dot(const uint64_t *buf, uint64_t width, uint64_t offset, unsigned dim, unsigned nbits, unsigned ext,
                  const uint64_t *L, uint64_t out[3]) {
    f3_t acc = f3_from(0);
    for (uint64_t k = 0; k < ((uint64_t)1 << nbits); ++k) {
        const uint64_t *c = buf + (k << ext) * width + offset;
        f3_t l = ld3(L + 3 * k);
        acc = f3_add(acc, dim == 1 ? f3_muls(l, c[0]) : f3_mul(ld3(c), l));
    }
    st3(out, acc);
}

/* stark_gen.rs:375-391: qq2[i][p*q_dim + k] = qq1[p*N + i][k] * (49^-N)^p  (qq1 = iNTT_Next(q)) */
void orc_qsplit(const uint64_t *qq1, unsigned nbits, unsigned q_dim, unsigned q_deg, uint64_t *qq2 /* [Next][q_dim*q_deg], zero-filled here */) {
    uint64_t N = (uint64_t)1 << nbits;
    uint64_t shift_inv = gl_pow(gl_inv(GL_SHIFT), N), cur = 1;
    for (unsigned p = 0; p < q_deg; ++p) {
        for (uint64_t i = 0; i < N; ++i)
            for (unsigned k = 0; k < q_dim; ++k)
                qq2[i * q_dim * q_deg + q_dim * p + k] = gl_mul(qq1[p * N * q_dim + i * q_dim + k], cur);
        cur = gl_mul(cur, shift_inv);
    }
}

/* stark_gen.rs:653-666 calculate_Z: z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1]; returns 1 iff it closes */
int orc_calculate_z(const uint64_t *num, const uint64_t *den, uint64_t n, uint64_t *z) {
    uint64_t *di = (uint64_t *)malloc(n * 24);
    orc_f3_batch_inverse(den, n, di);
    st3(z, f3_from(1));
    for (uint64_t i = 1; i < n; ++i) st3(z + 3 * i, f3_mul(ld3(z + 3 * (i - 1)), f3_mul(ld3(num + 3 * (i - 1)), ld3(di + 3 * (i - 1)))));
    f3_t chk = f3_mul(ld3(z + 3 * (n - 1)), f3_mul(ld3(num + 3 * (n - 1)), ld3(di + 3 * (n - 1))));
    free(di);
    return chk.v[0] == 1 && chk.v[1] == 0 && chk.v[2] == 0;
}

