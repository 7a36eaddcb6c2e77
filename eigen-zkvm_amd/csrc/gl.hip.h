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

// ZK-JIT-BEGIN  (the region up to ZK-JIT-END is also the prelude of the run-time compiled
// constraint kernels, csrc/expr_jit.hip; keep it free of #include and host-only code)
typedef unsigned long long u64;
typedef unsigned int u32;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL

namespace gl {

// add/sub are written on 32-bit limbs with explicit carries: on gfx950 v_add_co/v_addc_co are
// double-rate VALU ops while 64-bit compares + selects are not (tools/ubench_valu.hip,
// tools/field_probe.hip measured the variants: -20 % cycles on an NTT-shaped workload).
//
// GL_OPAQUE: every operation makes its operands opaque to the optimiser first.  Without it hipcc
// (ROCm 7.2, LLVM 22) folds a producer's plain add/sub into a consumer's carry op
// ("addcarry (add x, y), 0, cc -> addcarry x, y, cc") although the carry-OUT is used, which changes
// the carry and silently corrupts e.g. add(sub(a, b), 0) -- found by tests/test_gpu_stark_steps.py,
// reproduced in tools/dbg/dbg_f3b.hip.  The empty asm costs no instruction.
#define GL_OPAQUE(x) asm("" : "+v"(x))
__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ u64 add(u64 a, u64 b) {  // field_gl.rs:385-388; canonical in -> canonical out
    GL_OPAQUE(a); GL_OPAQUE(b);
    u32 c0, c1, d0, d1;
    u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
    u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
    u32 t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &d0);  // t = s - p = s + 2^32 - 1 (mod 2^64)
    u32 t1 = __builtin_addc(s1, 0u, d0, &d1);           // d1 <=> s >= p
    const bool sel = (c1 | d1) != 0;
    return mk64(sel ? t0 : s0, sel ? t1 : s1);
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {  // field_gl.rs:395-403
    GL_OPAQUE(a); GL_OPAQUE(b);
    u32 b0, b1, e0, e1;
    u32 d0 = __builtin_subc((u32)a, (u32)b, 0u, &b0);
    u32 d1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), b0, &b1);
    u32 m = 0u - b1;                                      // borrow ? 2^32 - 1 : 0;  d + p == d - (2^32 - 1)
    u32 r0 = __builtin_subc(d0, m, 0u, &e0);
    u32 r1 = __builtin_subc(d1, 0u, e0, &e1);
    return mk64(r0, r1);
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
// a*b mod p, any u64 inputs, canonical output (field_gl.rs:454-458, observable value).
//   product : four v_mad_u64_u32 (the compiler's a*b + __umul64hi(a,b) spends seven multiplies)
//   reduce  : x = lo + r2*2^64 + r3*2^96 = lo - r3 + r2*(2^32-1); the r2 term is ONE
//             v_mad_u64_u32 whose carry-out (vcc) selects the +2^32-1 fix-up.
// t + r2*(2^32-1) mod p, canonical: ONE v_mad_u64_u32 whose carry-out (vcc) selects the +2^32-1 fix-up
__device__ __forceinline__ u64 mad_eps(u32 r2, u64 t) {
    u64 u; u32 m;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, vcc"
                 : "=&v"(u), "=v"(m) : "v"(r2), "v"(t) : "vcc");
    u += m;                                              // carried: u += 2^32 - 1 (cannot carry again)
    u32 d0, d1;
    const u32 v0 = __builtin_addc((u32)u, 0xFFFFFFFFu, 0u, &d0);
    const u32 v1 = __builtin_addc((u32)(u >> 32), 0u, d0, &d1);  // d1 <=> u >= p
    return mk64(d1 ? v0 : (u32)u, d1 ? v1 : (u32)(u >> 32));
}
// lo + r2*2^64 + r3*2^96 mod p = lo - r3 + r2*(2^32-1), canonical; any 32-bit words
// (w1:w0) - r3, minus 2^32 - 1 more if that borrowed: five instructions with the borrows travelling as carry-ins.  Written as asm
// because the compiler lowers `subc(w1, 0, borrow)` to a v_cndmask + v_sub_co pair (one instruction more per field product: round 5);
// the wait states between a flag's producer and its consumer are the compiler's own (s_nop 1), which it cannot add inside asm.
__device__ __forceinline__ u64 sub_word_fold(u32 w0, u32 w1, u32 r3) {
    u32 t0, t1, mb;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\ts_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\ts_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(t0), "=&v"(t1), "=&v"(mb) : "v"(w0), "v"(w1), "v"(r3) : "vcc");
    return mk64(t0, t1);
}
__device__ __forceinline__ u64 reduce_words(u32 w0, u32 w1, u32 r2, u32 r3) { return mad_eps(r2, sub_word_fold(w0, w1, r3)); }
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
    GL_OPAQUE(a); GL_OPAQUE(b);
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    return reduce_words((u32)p0, (u32)p2, (u32)p3, (u32)(p3 >> 32));
}
// a*b + c mod p, any u64 inputs, canonical output: c rides on the product's multiply-adds
__device__ __forceinline__ u64 mul_add(u64 a, u64 b, u64 c) {
    GL_OPAQUE(a); GL_OPAQUE(b); GL_OPAQUE(c);
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0 + (u32)c;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32) + (c >> 32);   // <= (2^32-1)^2 + 2(2^32-1) = 2^64 - 1
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    return reduce_words((u32)p0, (u32)p2, (u32)p3, (u32)(p3 >> 32));
}
__device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }
__device__ __forceinline__ u64 pow(u64 a, u64 e) {  // field_gl.rs:467-479
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
    return r;
}
// a^(p - 2), p - 2 = 0xFFFFFFFE_FFFFFFFF = (2^31 - 1) 2^33 + (2^32 - 1): an addition chain of 64 squarings and 10 products (field_gl.rs:415-449
// walks its own chain of 72); square-and-multiply over the 63 one-bits took 125 -- an inversion is a single dependent chain wherever it occurs
__device__ __forceinline__ u64 sqr_n(u64 a, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) a = mul(a, a);
    return a;
}
__device__ __forceinline__ u64 inv(u64 a) {
    const u64 t2 = mul(mul(a, a), a);                 // a^(2^2 - 1)
    const u64 t3 = mul(mul(t2, t2), a);               // a^(2^3 - 1)
    const u64 t6 = mul(sqr_n(t3, 3), t3);
    const u64 t7 = mul(mul(t6, t6), a);
    const u64 t14 = mul(sqr_n(t7, 7), t7);
    const u64 t15 = mul(mul(t14, t14), a);
    const u64 t30 = mul(sqr_n(t15, 15), t15);
    const u64 t31 = mul(mul(t30, t30), a);            // a^(2^31 - 1)
    const u64 t32 = mul(mul(t31, t31), a);            // a^(2^32 - 1)
    return mul(sqr_n(t31, 33), t32);
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

// ZK-JIT-END

// ---- non-canonical ("nc") variants: results are SOME u64 congruent to the value, not necessarily < p.  Every
// operation here accepts such operands wherever it says "any u64"; chains of them (the Poseidon permutation)
// canonicalise once at the end instead of after every step (4-5 instructions per operation).
__device__ __forceinline__ u64 mad_eps_nc(u32 r2, u64 t) {          // t + r2*(2^32-1), any u64 representative
    u64 u; u32 m;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, vcc"
                 : "=&v"(u), "=v"(m) : "v"(r2), "v"(t) : "vcc");
    return u + m;                                                    // carried: += 2^32 - 1 (cannot carry again)
}
__device__ __forceinline__ u64 reduce_words_nc(u32 w0, u32 w1, u32 r2, u32 r3) { return mad_eps_nc(r2, sub_word_fold(w0, w1, r3)); }
// acc + x for a 32-bit word x as ONE multiply-add (x * 1 + acc) instead of a zero-extending move and a 64-bit addition
__device__ __forceinline__ u64 add_word(u64 acc, u32 x) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(carry) : "v"(x), "v"(acc));
    return d;
}
__device__ __forceinline__ u64 mul_nc(u64 a, u64 b) {                // any u64 in, nc out
    GL_OPAQUE(a); GL_OPAQUE(b);
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    return reduce_words_nc((u32)p0, (u32)p2, (u32)p3, (u32)(p3 >> 32));
}
// A squaring with the cross product a0 a1 taken once (round 4): three multiply-adds instead of four, the second use of the cross
// product a 64-bit addition.  The same number of instructions as the plain product, a cheaper mix: 2^22 x 19 tree 9.39-9.44 ms against
// 9.54-9.70 (profiles/r04/poseidon_variants.md).  Half of the S-box's products are squarings (x^2, x^6).
__device__ __forceinline__ u64 sqr_nc(u64 a) {
    GL_OPAQUE(a);
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
    const u64 p0 = (u64)a0 * a0;
    const u64 c = (u64)a0 * a1;
    const u64 p1 = c + (p0 >> 32);
    const u64 p2 = c + (u32)p1;
    const u64 p3 = (u64)a1 * a1 + (p1 >> 32) + (p2 >> 32);
    return reduce_words_nc((u32)p0, (u32)p2, (u32)p3, (u32)(p3 >> 32));
}
__device__ __forceinline__ u64 mul_add_nc(u64 a, u64 b, u64 c) {     // a*b + c, any u64 in, nc out
    GL_OPAQUE(a); GL_OPAQUE(b); GL_OPAQUE(c);
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p0 = (u64)a0 * b0 + (u32)c;
    const u64 p1 = (u64)a0 * b1 + (p0 >> 32) + (c >> 32);
    const u64 p2 = (u64)a1 * b0 + (u32)p1;
    const u64 p3 = (u64)a1 * b1 + (p1 >> 32) + (p2 >> 32);
    return reduce_words_nc((u32)p0, (u32)p2, (u32)p3, (u32)(p3 >> 32));
}
// a + b for any u64 a and CANONICAL b (< p): a + b - 2^64 <= p - 2, so the 2^32 - 1 fix-up cannot carry again
__device__ __forceinline__ u64 add_nc(u64 a, u64 b) {
    GL_OPAQUE(a); GL_OPAQUE(b);
    u32 c0, c1, d0, d1;
    const u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c0);
    const u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c0, &c1);
    const u32 m = 0u - c1;
    const u32 r0 = __builtin_addc(s0, m, 0u, &d0);
    const u32 r1 = __builtin_addc(s1, 0u, d0, &d1);
    return mk64(r0, r1);
}

// x * 2^E mod p for a compile-time 0 <= E < 96, canonical x -> canonical result.  2 has order 192
// and 2^96 = -1, so every root of unity of order <= 64 is a power of two (MG.0[6] = 2^39,
// constant.rs:54-68): the butterflies' small twiddles are shifts, not field multiplications.
// Callers fold E >= 96 into a swapped subtraction.
template <int E>
__device__ __forceinline__ u64 mul_pow2(u64 x) {
    static_assert(E >= 0 && E < 96, "fold 2^96 = -1 first");
    GL_OPAQUE(x);
    if constexpr (E == 0) {
        return x;
    } else if constexpr (E <= 32) {                       // lo + top*2^64, top < 2^32
        return mad_eps((u32)(x >> (64 - E)), x << E);
    } else if constexpr (E < 64) {                        // lo + hi*2^64, hi < 2^E
        const u64 lo = x << E, hi = x >> (64 - E);
        return reduce_words((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
    } else {                                              // E = 64 + r: t0*2^64 - (x*2^r >> 32), t0 = low word of x*2^r
        constexpr int r = E - 64;
        const u32 t0 = (u32)x << r;
        u64 m = ((u64)t0 << 32) - t0;                     // t0*(2^32-1) < p
        u64 s = r == 0 ? (x >> 32) : (x >> (32 - r));     // < p
        return sub(m, s);
    }
}

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

}  // namespace gl
