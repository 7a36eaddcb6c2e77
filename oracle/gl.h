/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
 * (eigen-zkvm_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Goldilocks field p = 2^64 - 2^32 + 1 in canonical (non-Montgomery) form.
 * Restates fields/src/field_gl.rs: MODULUS :12, add_assign :385-388, sub_assign :395-403,
 * mul_assign :454-458 (Montgomery there; every external artefact is the canonical as_int() :542,
 * so canonical u64 arithmetic yields the same observable values), inverse :415-449 (= a^(p-2)),
 * exp :467-479.  Cubic extension restates starky/src/f3g.rs (mul :407-449, _inv :207-235).
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

#define GL_P 0xFFFFFFFF00000001ULL

static inline uint64_t gl_red(uint64_t a) { return a >= GL_P ? a - GL_P : a; }
static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    __uint128_t s = (__uint128_t)a + b;
    if (s >= GL_P) s -= GL_P;
    return (uint64_t)s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
/* 128-bit product reduced with 2^64 = 2^32-1, 2^96 = -1 (mod p) -- the same reduction the
 * reference's packed path uses (fields/src/arch/x86_64/avx2_field_gl.rs:460 reduce128). */
static inline uint64_t gl_reduce128(__uint128_t x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & 0xFFFFFFFFULL;
    uint64_t t0, t2;
    if (__builtin_sub_overflow(lo, hi_hi, &t0)) t0 -= 0xFFFFFFFFULL;
    uint64_t t1 = hi_lo * 0xFFFFFFFFULL;
    if (__builtin_add_overflow(t0, t1, &t2)) t2 += 0xFFFFFFFFULL;
    return t2 >= GL_P ? t2 - GL_P : t2;
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((__uint128_t)a * b); }
static inline uint64_t gl_mul_slow(uint64_t a, uint64_t b) { return (uint64_t)(((__uint128_t)a * b) % GL_P); }
static inline uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

/* GF(p^3) = GF(p)[x]/(x^3 - x - 1), basis (1, x, x^2)  (f3g.rs:13-18; A.2 of SURVEY.md) */
typedef struct { uint64_t v[3]; } f3_t;
static inline f3_t f3_from(uint64_t a) { f3_t r = {{a, 0, 0}}; return r; }
static inline f3_t f3_add(f3_t a, f3_t b) { f3_t r = {{gl_add(a.v[0], b.v[0]), gl_add(a.v[1], b.v[1]), gl_add(a.v[2], b.v[2])}}; return r; }
static inline f3_t f3_sub(f3_t a, f3_t b) { f3_t r = {{gl_sub(a.v[0], b.v[0]), gl_sub(a.v[1], b.v[1]), gl_sub(a.v[2], b.v[2])}}; return r; }
static inline f3_t f3_muls(f3_t a, uint64_t s) { f3_t r = {{gl_mul(a.v[0], s), gl_mul(a.v[1], s), gl_mul(a.v[2], s)}}; return r; }
static inline f3_t f3_mul(f3_t a, f3_t b) { /* f3g.rs:420-430 */
    uint64_t A = gl_mul(gl_add(a.v[0], a.v[1]), gl_add(b.v[0], b.v[1]));
    uint64_t B = gl_mul(gl_add(a.v[0], a.v[2]), gl_add(b.v[0], b.v[2]));
    uint64_t C = gl_mul(gl_add(a.v[1], a.v[2]), gl_add(b.v[1], b.v[2]));
    uint64_t D = gl_mul(a.v[0], b.v[0]), E = gl_mul(a.v[1], b.v[1]), F = gl_mul(a.v[2], b.v[2]);
    uint64_t G = gl_sub(D, E);
    f3_t r;
    r.v[0] = gl_sub(gl_add(C, G), F);
    r.v[1] = gl_sub(gl_sub(gl_sub(gl_add(A, C), E), E), D);
    r.v[2] = gl_sub(B, G);
    return r;
}
static inline f3_t f3_inv(f3_t x) { /* f3g.rs:207-235 */
    uint64_t a = x.v[0], b = x.v[1], c = x.v[2];
    uint64_t aa = gl_mul(a, a), ac = gl_mul(a, c), ba = gl_mul(b, a), bb = gl_mul(b, b), bc = gl_mul(b, c), cc = gl_mul(c, c);
    uint64_t aaa = gl_mul(aa, a), aac = gl_mul(aa, c), abc = gl_mul(ba, c), abb = gl_mul(ba, b);
    uint64_t acc = gl_mul(ac, c), bbb = gl_mul(bb, b), bcc = gl_mul(bc, c), ccc = gl_mul(cc, c);
    uint64_t t = gl_neg(aaa);
    t = gl_sub(t, aac); t = gl_sub(t, aac);
    t = gl_add(t, abc); t = gl_add(t, abc); t = gl_add(t, abc);
    t = gl_add(t, abb); t = gl_sub(t, acc); t = gl_sub(t, bbb); t = gl_add(t, bcc); t = gl_sub(t, ccc);
    uint64_t ti = gl_inv(t);
    uint64_t i1 = gl_neg(aa);
    i1 = gl_sub(i1, ac); i1 = gl_sub(i1, ac); i1 = gl_add(i1, bc); i1 = gl_add(i1, bb); i1 = gl_sub(i1, cc);
    uint64_t i2 = gl_sub(ba, cc);
    uint64_t i3 = gl_add(gl_sub(ac, bb), cc);
    f3_t r = {{gl_mul(i1, ti), gl_mul(i2, ti), gl_mul(i3, ti)}};
    return r;
}
static inline f3_t f3_pow(f3_t a, uint64_t e) {
    f3_t r = f3_from(1);
    while (e) { if (e & 1) r = f3_mul(r, a); a = f3_mul(a, a); e >>= 1; }
    return r;
}

/* constant.rs:52-68: SHIFT = 49, MG.0[32] = 7^(2^32-1), MG.0[k] = MG.0[k+1]^2 */
#define GL_SHIFT 49ULL
static inline uint64_t gl_root(unsigned k) { /* primitive 2^k-th root of unity MG.0[k] */
    uint64_t w = gl_pow(7, 0xFFFFFFFFULL);
    for (unsigned n = 32; n > k; --n) w = gl_mul(w, w);
    return w;
}
#endif
