// Goldilocks field (p = 2^64 - 2^32 + 1) and its cubic extension for gfx950 device code.
//
// Representation: canonical u64 (NOT Montgomery).  The reference stores a*2^64 mod p
// (fields/src/field_gl.rs:16,454-458,525-538) but every observable artefact is the canonical
// as_int() (:542-544); its own SIMD path also computes in canonical form with the
// 2^64 = 2^32 - 1 reduction (fields/src/arch/x86_64/avx2_field_gl.rs:361,460), which is what
// maps best onto CDNA4: a 64x64->128 product is four v_mad_u64_u32 and the reduction is a
// handful of 32-bit add/sub-with-carry -- no second multiply as Montgomery would need.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL

namespace gl {

__host__ __device__ __forceinline__ u64 add(u64 a, u64 b) {  // field_gl.rs:385-388
    u64 s = a + b;
    return (s < a || s >= GL_P) ? s - GL_P : s;  // wrapped s - p == s + 2^32 - 1 (mod 2^64)
}
__host__ __device__ __forceinline__ u64 sub(u64 a, u64 b) {  // field_gl.rs:395-403
    u64 d = a - b;
    return a < b ? d + GL_P : d;
}
__host__ __device__ __forceinline__ u64 neg(u64 a) { return a ? GL_P - a : 0; }

// x = hi*2^64 + lo  ->  x mod p, canonical.  2^64 = 2^32 - 1, 2^96 = -1 (mod p).
__host__ __device__ __forceinline__ u64 reduce128(u64 lo, u64 hi) {
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += GL_EPS;
    return t2 >= GL_P ? t2 - GL_P : t2;
}
__device__ __forceinline__ u64 mul(u64 a, u64 b) {  // field_gl.rs:454-458 (observable value)
    return reduce128(a * b, __umul64hi(a, b));
}
__device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }
__device__ __forceinline__ u64 pow(u64 a, u64 e) {  // field_gl.rs:467-479
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
    return r;
}
__device__ __forceinline__ u64 inv(u64 a) { return pow(a, GL_P - 2); }  // field_gl.rs:415-449

// host-side twins (used only to build twiddle tables at context creation)
inline u64 hmul(u64 a, u64 b) {
    unsigned __int128 x = (unsigned __int128)a * b;
    return reduce128((u64)x, (u64)(x >> 64));
}
inline u64 hpow(u64 a, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = hmul(r, a); a = hmul(a, a); e >>= 1; }
    return r;
}
inline u64 hinv(u64 a) { return hpow(a, GL_P - 2); }
inline u64 hroot(unsigned k) {  // MG.0[k], starky/src/constant.rs:54-68
    u64 w = hpow(7, 0xFFFFFFFFULL);
    for (unsigned n = 32; n > k; --n) w = hmul(w, w);
    return w;
}

// GF(p^3) = GF(p)[x]/(x^3 - x - 1)  (starky/src/f3g.rs).  Device values never carry the
// reference's runtime `dim` tag: the width (1 or 3 words) is a static property of each buffer.
struct f3 { u64 v[3]; };
__device__ __forceinline__ f3 f3_add(f3 a, f3 b) { return f3{{add(a.v[0], b.v[0]), add(a.v[1], b.v[1]), add(a.v[2], b.v[2])}}; }
__device__ __forceinline__ f3 f3_sub(f3 a, f3 b) { return f3{{sub(a.v[0], b.v[0]), sub(a.v[1], b.v[1]), sub(a.v[2], b.v[2])}}; }
__device__ __forceinline__ f3 f3_muls(f3 a, u64 s) { return f3{{mul(a.v[0], s), mul(a.v[1], s), mul(a.v[2], s)}}; }
__device__ __forceinline__ f3 f3_mul(f3 a, f3 b) {  // f3g.rs:420-430
    u64 A = mul(add(a.v[0], a.v[1]), add(b.v[0], b.v[1]));
    u64 B = mul(add(a.v[0], a.v[2]), add(b.v[0], b.v[2]));
    u64 C = mul(add(a.v[1], a.v[2]), add(b.v[1], b.v[2]));
    u64 D = mul(a.v[0], b.v[0]), E = mul(a.v[1], b.v[1]), F = mul(a.v[2], b.v[2]);
    u64 G = sub(D, E);
    return f3{{sub(add(C, G), F), sub(sub(sub(add(A, C), E), E), D), sub(B, G)}};
}
__device__ __forceinline__ f3 f3_inv(f3 x) {  // f3g.rs:207-235
    u64 a = x.v[0], b = x.v[1], c = x.v[2];
    u64 aa = mul(a, a), ac = mul(a, c), ba = mul(b, a), bb = mul(b, b), bc = mul(b, c), cc = mul(c, c);
    u64 aaa = mul(aa, a), aac = mul(aa, c), abc = mul(ba, c), abb = mul(ba, b);
    u64 acc = mul(ac, c), bbb = mul(bb, b), bcc = mul(bc, c), ccc = mul(cc, c);
    u64 t = neg(aaa);
    t = sub(t, aac); t = sub(t, aac);
    t = add(t, abc); t = add(t, abc); t = add(t, abc);
    t = add(t, abb); t = sub(t, acc); t = sub(t, bbb); t = add(t, bcc); t = sub(t, ccc);
    u64 ti = inv(t);
    u64 i1 = neg(aa);
    i1 = sub(i1, ac); i1 = sub(i1, ac); i1 = add(i1, bc); i1 = add(i1, bb); i1 = sub(i1, cc);
    u64 i2 = sub(ba, cc);
    u64 i3 = add(sub(ac, bb), cc);
    return f3{{mul(i1, ti), mul(i2, ti), mul(i3, ti)}};
}

}  // namespace gl
