// The dense T x T layers of the scalar-field Poseidon (poseidon_bn128_opt.rs:121-131, 190-199: the MDS matrix M of the full rounds and
// the pre-sparse matrix P) on the matrix pipe, one permutation per lane (round 6; the Goldilocks form of the same idea: gl_mfma.hip.h).
// Included by frhash_impl.hip.h inside the field's namespace, after fe29_impl.hip.h (NR, LB, LMASK, Q29, QINV29, fe).  No include guard.
//
// out[o] = sum_j c[o][j] x[j] mod r, x in the internal Montgomery form (9 x 29-bit limbs, any value < 2^256), c canonical constants:
//   * x[j] is 32 bytes x_b; c x = sum_b x_b (c 2^(8 b) 2^29 mod r), each constant written in 32 balanced base-256 digits (its
//     representative in (-r/2, r/2] fits): the A operand of v_mfma_i32_32x32x32_i8 for (output o, input j) is the 32 x 32 digit block
//     [digit d][byte b], the B operand the 64 lanes' bytes of word j (x_b xor 0x80 = x_b - 128; the missing 128 sum(a) is a constant per
//     output).  One tile = one output, one k-step = one input word: T^2 x 2 MFMAs per layer for the wave's 64 permutations, T^2 KB of
//     fragments per matrix (289 KB at t = 17: global memory, read through the caches by every wave in the same order).
//   * the 32 i32 digit columns of an output (|column| < 2^23: checked when the table is built) become eight biased 64-bit words of weight
//     2^(32 q) (two v_lshl_add + two v_mad_i64_i32 each; the round constants' image M c, the 128 sum(a) and the bias ride on the eight
//     addends K), those 274 bits become ten 29-bit limbs, and ONE Montgomery step (the 2^29 in the table) brings the value below
//     2^245 + r < 2r: ~110 vector instructions per output against the 17 products (~1 800 multiply-adds) of the vector-pipe row.
//   * lanes l and l + 32 trade halves by v_permlane32_swap on the way in (each supplies 16 of a word's 32 bytes for both permutations) and
//     on the way out (each forms the low / high four words of both outputs).
// Every lane of the wave must take part.

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

// ---- host: 256-bit residues as four u64, only what building a table needs --------------------------------------------------------
struct MfInt { u64 w[5]; };   // two's complement, 320 bits
inline MfInt mf_zero() { return MfInt{{0, 0, 0, 0, 0}}; }
inline MfInt mf_add(const MfInt& a, const MfInt& b) {
    MfInt r; unsigned __int128 c = 0;
    for (int i = 0; i < 5; ++i) { c += (unsigned __int128)a.w[i] + b.w[i]; r.w[i] = (u64)c; c >>= 64; }
    return r;
}
inline MfInt mf_neg(const MfInt& a) {
    MfInt r; unsigned __int128 c = 1;
    for (int i = 0; i < 5; ++i) { c += (unsigned __int128)(~a.w[i]); r.w[i] = (u64)c; c >>= 64; }
    return r;
}
inline MfInt mf_sub(const MfInt& a, const MfInt& b) { return mf_add(a, mf_neg(b)); }
inline bool mf_negative(const MfInt& a) { return (a.w[4] >> 63) != 0; }
inline bool mf_less(const MfInt& a, const MfInt& b) { return mf_negative(mf_sub(a, b)); }          // both of magnitude < 2^318
inline MfInt mf_modulus() {
    MfInt r = mf_zero();
    for (int k = 0; k < NR; ++k) {
        const int bit = LB * k;
        r.w[bit >> 6] |= (u64)Q29(k) << (bit & 63);
        if ((bit & 63) + LB > 64) r.w[(bit >> 6) + 1] |= (u64)Q29(k) >> (64 - (bit & 63));
    }
    return r;
}
inline MfInt mf_addmod(const MfInt& a, const MfInt& b, const MfInt& q) { MfInt s = mf_add(a, b); return mf_less(s, q) ? s : mf_sub(s, q); }
inline MfInt mf_dblmod(const MfInt& a, const MfInt& q) { return mf_addmod(a, a, q); }
// t mod q for 0 <= t < 2^40 q: the quotient estimated from the top 128 bits (never too large, at most a few too small)
inline MfInt mf_reduce_small(MfInt t, const MfInt& q) {
    const unsigned __int128 th = ((unsigned __int128)t.w[4] << 64) | t.w[3];
    const u64 k = (u64)(th / ((unsigned __int128)q.w[3] + 1));
    MfInt kq = mf_zero(); unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)q.w[i] * k; kq.w[i] = (u64)c; c >>= 64; }
    kq.w[4] = (u64)c;
    t = mf_sub(t, kq);
    while (!mf_less(t, q)) t = mf_sub(t, q);
    return t;
}
inline MfInt mf_mulsmall(const MfInt& a, u64 k, const MfInt& q) {                                   // a k mod q, a < q, k < 2^40
    MfInt t = mf_zero(); unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.w[i] * k; t.w[i] = (u64)c; c >>= 64; }
    t.w[4] = (u64)c;
    return mf_reduce_small(t, q);
}
inline MfInt mf_shlmod(MfInt a, int k, const MfInt& q) {                                            // a 2^k mod q
    for (; k >= 32; k -= 32) a = mf_mulsmall(a, 1ull << 32, q);
    return k ? mf_mulsmall(a, 1ull << k, q) : a;
}
inline MfInt mf_mulmod(const MfInt& a, const MfInt& b, const MfInt& q) {                            // a, b < q: Horner over b's 32-bit words
    MfInt r = mf_zero();
    for (int i = 7; i >= 0; --i) {
        r = mf_mulsmall(r, 1ull << 32, q);
        r = mf_addmod(r, mf_mulsmall(a, (u32)(b.w[i >> 1] >> (32 * (i & 1))), q), q);
    }
    return r;
}
inline MfInt mf_from_bytes(const unsigned char* p) {                                                 // 32 bytes, little endian, < q
    MfInt r = mf_zero();
    for (int i = 0; i < 32; ++i) r.w[i >> 3] |= (u64)p[i] << (8 * (i & 7));
    return r;
}
// v in [0, q) -> the 32 balanced digits of its representative in (-q/2, q/2]; false if they do not spell it
inline bool mf_digits(const MfInt& v, const MfInt& q, signed char* dg) {
    MfInt s = v;
    if (mf_less(mf_sub(q, v), v)) s = mf_sub(v, q);                                                  // q - v < v: take v - q
    int carry = 0;
    for (int d = 0; d < 32; ++d) {                                                                   // two's complement bytes, carry = 1 after a digit taken negative
        const int t = (int)((s.w[d >> 3] >> (8 * (d & 7))) & 0xFF) + carry;
        carry = t >= 128;
        dg[d] = (signed char)(carry ? t - 256 : t);
    }
    return carry ? s.w[4] == ~0ull : s.w[4] == 0;                                                    // what is left: the sign bytes + carry = 0
}

// One matrix: frag [T][T][64 lanes][16 bytes]; corr[o] = 128 * sum of the row's constants (mod r).  mat: T * T canonical 32-byte
// integers, out[o] = sum_j mat[j * T + o] x[j] (the layout of the permutation's tables).  Returns "" or what is wrong.
inline std::string mf_build_matrix(const unsigned char* mat, int T, signed char* frag, MfInt* corr) {
    const MfInt q = mf_modulus();
    std::vector<signed char> dig((size_t)32 * 32);
    for (int o = 0; o < T; ++o) {
        corr[o] = mf_zero();
        long long colabs[32] = {0};
        for (int j = 0; j < T; ++j) {
            MfInt v = mf_shlmod(mf_from_bytes(mat + 32 * ((size_t)j * T + o)), LB, q);            // c 2^29: one Montgomery step follows
            for (int b = 0; b < 32; ++b) {
                if (!mf_digits(v, q, &dig[(size_t)b * 32])) return "a constant does not fit 32 balanced digits";
                for (int d = 0; d < 32; ++d) colabs[d] += dig[(size_t)b * 32 + d] < 0 ? -dig[(size_t)b * 32 + d] : dig[(size_t)b * 32 + d];
                corr[o] = mf_addmod(corr[o], v, q);
                v = mf_shlmod(v, 8, q);
            }
            // fragment (o, j): lane l -> row m = l & 31 = digit 16 H + idx, K half h = l >> 5 = bytes 16 h .. 16 h + 15
            for (int l = 0; l < 64; ++l) {
                const int m = l & 31, H = (m >> 2) & 1, idx = (m & 3) + 4 * (m >> 3), d = 16 * H + idx, h = l >> 5;
                for (int i = 0; i < 16; ++i) frag[(((size_t)o * T + j) * 64 + l) * 16 + i] = dig[(size_t)(16 * h + i) * 32 + d];
            }
        }
        for (int d = 0; d < 32; ++d) if (128 * colabs[d] >= (1ll << 23) - (1ll << 15)) return "a digit column could leave the range of the recombination";
        corr[o] = mf_shlmod(corr[o], 7, q);                                                         // 128 x the sum of the row's constants
    }
    return "";
}
// The eight addends of output o of one layer: words of (add + corr - bias) mod r on top of 2^48 each, bias = sum_q 2^48 2^(32 q).
inline void mf_addends(const MfInt& add /* < r: 2^29 (M c) in the internal form, or 0 */, const MfInt& corr, u64* K) {
    const MfInt q = mf_modulus();
    MfInt bias = mf_zero(), one = mf_zero(); one.w[0] = 1;
    for (int w = 0; w < 8; ++w) bias = mf_addmod(bias, mf_shlmod(one, 48 + 32 * w, q), q);
    MfInt k = mf_addmod(add, corr, q);
    k = mf_addmod(k, mf_sub(q, bias), q);
    for (int w = 0; w < 8; ++w) K[w] = (1ull << 48) + (u32)(k.w[w >> 1] >> (32 * (w & 1)));
}

// What the device computes for output o from a table, step by step in host integers (the i32 columns, the eight biased words, the ten
// limbs, the Montgomery step): x[j] any integers < 2^256.  With mf_selfcheck() and the known answers on the device this pins the form on
// both sides.  ok = false if a column or a word leaves the range the device code assumes.
inline u64 mf_neg_inverse() {                                             // -r^-1 mod 2^64
    const u64 r0 = mf_modulus().w[0];
    u64 inv = 1;
    for (int i = 0; i < 7; ++i) inv *= 2 - r0 * inv;
    return (u64)0 - inv;
}
inline MfInt mf_emulate_gen(const signed char* const* frags /* per input word: its fragment for this output */, int n_in, const u64* K8, const MfInt* x,
                            int shift /* the Montgomery step: 29 or 32 */, bool& ok) {
    long long col[32];
    for (int d = 0; d < 32; ++d) {
        const int H = d >> 4, idx = d & 15, m = (idx & 3) + 8 * (idx >> 2) + 4 * H;
        long long c = 0;
        for (int j = 0; j < n_in; ++j)
            for (int b = 0; b < 32; ++b) {
                const int h = b >> 4, i = b & 15;
                const int a = frags[j][(size_t)(32 * h + m) * 16 + i];
                const int xb = (int)(signed char)((unsigned char)(x[j].w[b >> 3] >> (8 * (b & 7))) ^ 0x80);
                c += (long long)a * xb;
            }
        col[d] = c;
        if (c >= (1ll << 23) || c <= -(1ll << 23)) ok = false;
    }
    MfInt V = mf_zero();
    for (int q = 7; q >= 0; --q) {
        const long long w = (long long)K8[q] + col[4 * q] + (col[4 * q + 1] << 8) + (col[4 * q + 2] << 16) + (col[4 * q + 3] << 24);
        if (w <= 0 || w >= (1ll << 49)) ok = false;
        for (int i = 4; i > 0; --i) V.w[i] = (V.w[i] << 32) | (V.w[i - 1] >> 32);               // V = V 2^32 + w
        V.w[0] <<= 32;
        MfInt ww = mf_zero(); ww.w[0] = (u64)w;
        V = mf_add(V, ww);
    }
    const u64 mask = (1ull << shift) - 1, m = (V.w[0] * mf_neg_inverse()) & mask;
    MfInt mr = mf_zero();
    const MfInt q = mf_modulus();
    for (int bit = shift - 1; bit >= 0; --bit) { mr = mf_add(mr, mr); if ((m >> bit) & 1) mr = mf_add(mr, q); }
    V = mf_add(V, mr);
    if (V.w[0] & mask) ok = false;
    for (int i = 0; i < 5; ++i) V.w[i] = (V.w[i] >> shift) | (i + 1 < 5 ? V.w[i + 1] << (64 - shift) : 0);
    return V;
}
inline MfInt mf_emulate(const signed char* frag, const u64* K8, int T, int o, const MfInt* x, bool& ok) {
    std::vector<const signed char*> f(T);
    for (int j = 0; j < T; ++j) f[j] = frag + ((size_t)o * T + j) * 1024;
    return mf_emulate_gen(f.data(), T, K8, x, LB, ok);
}
// Host-only check of the tables of one matrix (no GPU): built from `mat`, every output of random and extreme vectors, emulated the
// device's way, must be congruent to sum_j c x_j + A for a random addend A, below 2r, with every intermediate in range.
inline std::string mf_selfcheck(const unsigned char* mat, int T, u64 seed) {
    const MfInt q = mf_modulus();
    std::vector<signed char> frag((size_t)T * T * 1024);
    std::vector<MfInt> corr(T);
    const std::string err = mf_build_matrix(mat, T, frag.data(), corr.data());
    if (!err.empty()) return err;
    auto rnd = [&]() { seed += 0x9E3779B97F4A7C15ull; u64 z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    auto reduce = [&](MfInt v) { while (!mf_less(v, q)) v = mf_sub(v, q); return v; };
    std::vector<MfInt> x(T);
    for (int trial = 0; trial < 5; ++trial) {
        for (int j = 0; j < T; ++j) {
            x[j] = mf_zero();
            for (int i = 0; i < 4; ++i) x[j].w[i] = trial == 0 ? ~0ull : trial == 1 ? 0 : trial == 2 ? 0x8080808080808080ull : trial == 3 ? 0x7F7F7F7F7F7F7F7Full : rnd();
        }
        for (int o = 0; o < T; ++o) {
            MfInt A = mf_zero(); for (int i = 0; i < 4; ++i) A.w[i] = rnd(); A.w[3] >>= 3; A = reduce(A);
            u64 K8[8];
            mf_addends(mf_shlmod(A, LB, q), corr[o], K8);
            bool ok = true;
            const MfInt R = mf_emulate(frag.data(), K8, T, o, x.data(), ok);
            MfInt want = A;
            for (int j = 0; j < T; ++j) want = mf_addmod(want, mf_mulmod(reduce(x[j]), mf_from_bytes(mat + 32 * ((size_t)j * T + o)), q), q);
            if (!ok) return "t = " + std::to_string(T) + ", output " + std::to_string(o) + ": an intermediate leaves its range";
            if (!mf_less(R, mf_add(q, q))) return "t = " + std::to_string(T) + ", output " + std::to_string(o) + ": result not below 2r";
            const MfInt Rr = reduce(R);
            for (int i = 0; i < 5; ++i) if (Rr.w[i] != want.w[i]) return "t = " + std::to_string(T) + ", output " + std::to_string(o) + ": not congruent to the product";
        }
    }
    return "";
}

// ---- the sparse rounds (poseidon_bn128_opt.rs:133-188) ------------------------------------------------------------------------------
// Round r: y = x_0^5 + c_r;  x_0' = sum_j S_r[j] x_j (x_0 = y);  x_k' = x_k + S'_r[k] y.  Every one of these is a product of the matrix
// pipe's kind: the row against all t words, column k against (x_0^5, x_k) with the coefficient 1 for x_k -- so the running words never
// leave the byte form the B operands want, come back below 2^241 + r every round (no renormalisation schedule), and the vector pipe is
// left with the S-box and ~80 instructions per word and round.  Montgomery step 2^32 here (the outputs are wanted as eight 32-bit words).
// Per round, contiguous: 2t - 1 fragments [row j = 0..t-1 | column k = 1..t-1], then t x 8 addends (c_r's images and the biases).
constexpr int MF_SP_SHIFT = 32;
inline size_t mf_sparse_round_bytes(int T) { return (size_t)(2 * T - 1) * 1024 + (size_t)T * 64; }
// one fragment from a coefficient already multiplied by the step's power of two; adds to the output's corr and column sums
inline bool mf_fragment(MfInt v, const MfInt& q, signed char* frag, MfInt& corr, long long* colabs) {
    signed char dig[32 * 32];
    for (int b = 0; b < 32; ++b) {
        if (!mf_digits(v, q, dig + b * 32)) return false;
        for (int d = 0; d < 32; ++d) colabs[d] += dig[b * 32 + d] < 0 ? -dig[b * 32 + d] : dig[b * 32 + d];
        corr = mf_addmod(corr, mf_shlmod(v, 7, q), q);
        v = mf_shlmod(v, 8, q);
    }
    for (int l = 0; l < 64; ++l) {
        const int m = l & 31, H = (m >> 2) & 1, idx = (m & 3) + 4 * (m >> 3), d = 16 * H + idx, h = l >> 5;
        for (int i = 0; i < 16; ++i) frag[(size_t)l * 16 + i] = dig[(16 * h + i) * 32 + d];
    }
    return true;
}
// S: n_rp x (2t - 1) canonical 32-byte integers (row t | column t - 1), c: the n_rp round constants; out: n_rp blocks of
// mf_sparse_round_bytes(t), ident: the fragment of the coefficient 1.  "" or what is wrong.
inline std::string mf_build_sparse(const unsigned char* S, const unsigned char* c, int T, int n_rp, unsigned char* out, signed char* ident) {
    const MfInt q = mf_modulus();
    MfInt one = mf_zero(); one.w[0] = 1;
    MfInt corr_i = mf_zero(); long long abs_i[32] = {0};
    if (!mf_fragment(mf_shlmod(one, MF_SP_SHIFT, q), q, ident, corr_i, abs_i)) return "the unit does not fit 32 balanced digits";
    const size_t rb = mf_sparse_round_bytes(T);
    for (int r = 0; r < n_rp; ++r) {
        unsigned char* blk = out + rb * r;
        u64* K = reinterpret_cast<u64*>(blk + (size_t)(2 * T - 1) * 1024);
        const unsigned char* Sr = S + 32 * (size_t)(2 * T - 1) * r;
        const MfInt cr = mf_shlmod(mf_from_bytes(c + 32 * (size_t)r), NR * LB + MF_SP_SHIFT, q);   // c_r in the internal form, times the step's 2^32
        {   // the row
            MfInt corr = mf_zero(); long long ab[32] = {0};
            for (int j = 0; j < T; ++j)
                if (!mf_fragment(mf_shlmod(mf_from_bytes(Sr + 32 * j), MF_SP_SHIFT, q), q, (signed char*)blk + (size_t)j * 1024, corr, ab)) return "a constant does not fit 32 balanced digits";
            for (int d = 0; d < 32; ++d) if (128 * ab[d] >= (1ll << 23) - (1ll << 15)) return "a digit column could leave the range of the recombination";
            mf_addends(mf_mulmod(cr, mf_from_bytes(Sr), q), corr, K);
        }
        for (int k = 1; k < T; ++k) {
            MfInt corr = corr_i; long long ab[32];
            for (int d = 0; d < 32; ++d) ab[d] = abs_i[d];
            const MfInt sk = mf_from_bytes(Sr + 32 * (size_t)(T + k - 1));
            if (!mf_fragment(mf_shlmod(sk, MF_SP_SHIFT, q), q, (signed char*)blk + (size_t)(T + k - 1) * 1024, corr, ab)) return "a constant does not fit 32 balanced digits";
            for (int d = 0; d < 32; ++d) if (128 * ab[d] >= (1ll << 23) - (1ll << 15)) return "a digit column could leave the range of the recombination";
            mf_addends(mf_mulmod(cr, sk, q), corr, K + 8 * k);
        }
    }
    return "";
}
// Host-only check of the sparse rounds' tables: a few rounds replayed the device's way on random and extreme words against plain arithmetic.
inline std::string mf_selfcheck_sparse(const unsigned char* S, const unsigned char* c, int T, int n_rp, u64 seed) {
    const MfInt q = mf_modulus();
    const size_t rb = mf_sparse_round_bytes(T);
    std::vector<unsigned char> tab(rb * n_rp);
    std::vector<signed char> ident(1024);
    const std::string err = mf_build_sparse(S, c, T, n_rp, tab.data(), ident.data());
    if (!err.empty()) return err;
    auto rnd = [&]() { seed += 0x9E3779B97F4A7C15ull; u64 z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    auto reduce = [&](MfInt v) { while (!mf_less(v, q)) v = mf_sub(v, q); return v; };
    std::vector<MfInt> x(T);
    const int rounds[4] = {0, 1, n_rp / 2, n_rp - 1};
    for (int t4 = 0; t4 < 4; ++t4) {
        const int r = rounds[t4];
        for (int j = 0; j < T; ++j) { x[j] = mf_zero(); for (int i = 0; i < 4; ++i) x[j].w[i] = t4 == 0 ? ~0ull : t4 == 1 ? 0x8080808080808080ull >> (j & 1) : rnd(); }
        const unsigned char* blk = tab.data() + rb * r;
        const u64* K = reinterpret_cast<const u64*>(blk + (size_t)(2 * T - 1) * 1024);
        const unsigned char* Sr = S + 32 * (size_t)(2 * T - 1) * r;
        const MfInt cr = mf_shlmod(mf_from_bytes(c + 32 * (size_t)r), NR * LB, q);                 // internal form
        const MfInt y = mf_addmod(reduce(x[0]), cr, q);                                             // x[0] plays x_0^5
        for (int o = 0; o < T; ++o) {
            bool ok = true;
            MfInt R, want;
            if (o == 0) {
                std::vector<const signed char*> f(T);
                for (int j = 0; j < T; ++j) f[j] = (const signed char*)blk + (size_t)j * 1024;
                R = mf_emulate_gen(f.data(), T, K, x.data(), MF_SP_SHIFT, ok);
                want = mf_mulmod(y, mf_from_bytes(Sr), q);
                for (int j = 1; j < T; ++j) want = mf_addmod(want, mf_mulmod(reduce(x[j]), mf_from_bytes(Sr + 32 * j), q), q);
            } else {
                const signed char* f[2] = {(const signed char*)blk + (size_t)(T + o - 1) * 1024, ident.data()};
                const MfInt xx[2] = {x[0], x[o]};
                R = mf_emulate_gen(f, 2, K + 8 * o, xx, MF_SP_SHIFT, ok);
                want = mf_addmod(reduce(x[o]), mf_mulmod(y, mf_from_bytes(Sr + 32 * (size_t)(T + o - 1)), q), q);
            }
            const std::string at = "sparse round " + std::to_string(r) + " of t = " + std::to_string(T) + ", word " + std::to_string(o);
            if (!ok) return at + ": an intermediate leaves its range";
            if (R.w[4] || (R.w[3] >> 63) || !mf_less(R, mf_add(q, q))) return at + ": result not below 2r";
            const MfInt Rr = reduce(R);
            for (int i = 0; i < 5; ++i) if (Rr.w[i] != want.w[i]) return at + ": not congruent to the round's arithmetic";
        }
    }
    return "";
}

// ---- device ----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mf_mad_i64(int a, int b, u64 c) {          // a * b + c, signed 32 x 32 + 64
    u64 d, carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ void mf_swap32(u32& a, u32& b) {                // a's lanes 32..63 <-> b's lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);   // (as inline assembly the swap escaped the compiler's hazard handling: wrong digests at t = 17)
    a = r[0]; b = r[1];
}
// the 32 bytes of x (< 2^256, limbs normalised) as this lane's share of the two B operands
__device__ __forceinline__ void mf_make_b(const fe& x, mf_v4i& b0, mf_v4i& b1) {
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int bit = 32 * i, k = bit / LB, s = bit % LB;
        u32 v = x.l[k] >> s;
        if (k + 1 < NR) v |= x.l[k + 1] << (LB - s);
        w[i] = v ^ 0x80808080u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) mf_swap32(w[i], w[4 + i]);                 // w[0..3]: lanes < 32 their own low half, lanes >= 32 the partner's high half
    b0 = mf_v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};               // column tile 0 = the permutations of lanes 0..31
    b1 = mf_v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};               // column tile 1 = those of lanes 32..63
}
// sixteen digit columns (this lane's half of an output) + their four addends -> four words of weight 2^(32 g), each in (2^47, 2^49)
__device__ __forceinline__ void mf_words(const mf_v16i& c, const u64* __restrict__ K, int one, int s16, u64 (&W)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int e0 = (c[4 * g + 1] << 8) + c[4 * g], e1 = (c[4 * g + 3] << 8) + c[4 * g + 2];   // |e| < 2^31
        W[g] = mf_mad_i64(e1, s16, mf_mad_i64(e0, one, K[g]));
    }
}
// eight words of weight 2^(32 q), each < 2^49, holding 2^29 * (the output) -> the output, < 2^245 + r, limbs normalised
__device__ __forceinline__ fe mf_reduce(const u64 (&W)[8]) {
    u32 v[10];
    u64 acc = W[0];
    v[0] = (u32)acc;
#pragma unroll
    for (int q = 1; q < 8; ++q) { acc = W[q] + (acc >> 32); v[q] = (u32)acc; }
    v[8] = (u32)(acc >> 32); v[9] = 0;
    u64 t[NR + 1];
#pragma unroll
    for (int k = 0; k <= NR; ++k) {
        const int bit = LB * k, wi = bit >> 5, s = bit & 31;
        t[k] = (s ? __builtin_amdgcn_alignbit(v[wi + 1], v[wi], s) : v[wi]) & LMASK;
    }
    const u32 m = ((u32)t[0] * QINV29) & LMASK;
#pragma unroll
    for (int j = 0; j < NR; ++j) t[j] += (u64)m * Q29(j);
    t[1] += t[0] >> LB;                                                    // the low 29 bits of t[0] are zero now
    fe r;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (k + 1 < NR) { r.l[k] = (u32)t[k + 1] & LMASK; t[k + 2] += t[k + 1] >> LB; }
        else r.l[k] = (u32)t[k + 1];
    }
    return r;
}
// st <- the layer's image of st (every word < 2^256 in, < 2r out).  frag: the matrix's fragments (global memory), K: the layer's addends
// [T][8], abuf: 2 x T x 64 fragments' worth of LDS.  The WHOLE BLOCK must be here: its threads fetch an output's T fragments together, one
// output ahead of the products (a fragment read straight from global memory by the wave that needs it cost its full latency 289 times a
// layer: the products of t = 17 ran at a tenth of the pipe's rate).
template <int T>
__device__ __forceinline__ void mf_dense(fe (&st)[T], const mf_v4i* __restrict__ frag, const u64* __restrict__ K, mf_v4i* abuf) {
    constexpr int PER = T * 64, NLD = (PER + 255) / 256;                   // fragments' 16-byte words per output; loads per thread (block of 256)
    const int lane = threadIdx.x & 63, H = lane >> 5;
    mf_v4i pf[NLD];
    auto fetch = [&](int o) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int e = threadIdx.x + 256 * k; if (NLD * 256 == PER || e < PER) pf[k] = frag[o * PER + e]; }
    };
    auto stash = [&](int o) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int e = threadIdx.x + 256 * k; if (NLD * 256 == PER || e < PER) abuf[(o & 1) * PER + e] = pf[k]; }
    };
    fetch(0);
    mf_v4i B0[T], B1[T];
    fh_static_for<0, T>([&](auto J) { constexpr int j = decltype(J)::value; mf_make_b(st[j], B0[j], B1[j]); });
    stash(0);
    __syncthreads();
    int one = 1, s16 = 65536;
    asm volatile("" : "+v"(one), "+v"(s16));
    fh_static_for<0, T>([&](auto O) {
        constexpr int o = decltype(O)::value;
        if constexpr (o + 1 < T) fetch(o + 1);
        const mf_v4i* __restrict__ fr = abuf + (o & 1) * PER + lane;
        mf_v16i acc0 = {}, acc1 = {};
        fh_static_for<0, T>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const mf_v4i a = fr[j * 64];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[j], acc1, 0, 0, 0);
        });
        const u64* __restrict__ Ko = K + 8 * o + 4 * H;
        u64 Wa[4], Wb[4];
        mf_words(acc0, Ko, one, s16, Wa);                                  // tile 0: lanes < 32 words 0..3 of their own output, lanes >= 32 words 4..7 of the partner's
        mf_words(acc1, Ko, one, s16, Wb);                                  // tile 1: lanes < 32 words 0..3 of the partner's,  lanes >= 32 words 4..7 of their own
        u64 W[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32 a0 = (u32)Wa[g], a1 = (u32)(Wa[g] >> 32), b0 = (u32)Wb[g], b1 = (u32)(Wb[g] >> 32);
            mf_swap32(a0, b0); mf_swap32(a1, b1);                          // now a = the lane's own words 0..3, b = its own words 4..7, in both halves
            W[g] = ((u64)a1 << 32) | a0; W[4 + g] = ((u64)b1 << 32) | b0;
        }
        st[o] = mf_reduce(W);
        if constexpr (o + 1 < T) { stash(o + 1); __syncthreads(); }        // the other buffer: its last readers passed the previous barrier
    });
    __syncthreads();                                                       // the next layer's first stash must not overtake this layer's last reads
}

// ---- device: the sparse rounds -------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr u32 mf_r32(int i) {                          // the modulus in 32-bit words
    u64 acc = 0;
    for (int k = 0; k < NR; ++k) {
        const int bit = LB * k - 32 * i;
        if (bit >= 0 && bit < 32) acc |= (u64)Q29(k) << bit;
        else if (bit < 0 && bit > -LB) acc |= (u64)Q29(k) >> (-bit);
    }
    return (u32)acc;
}
__host__ __device__ constexpr u32 mf_rinv32() {                            // -r^-1 mod 2^32
    u32 inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - mf_r32(0) * inv;
    return 0u - inv;
}
// eight words of weight 2^(32 q), each < 2^49 (lo | hi halves), holding 2^32 * (the output) -> the output in eight 32-bit words, < 2^242 + r
__device__ __forceinline__ void mf_reduce_words(const u32 (&lo)[8], const u32 (&hi)[8], u32 (&out)[8]) {
    u32 v[9];
    v[0] = lo[0];
    u32 c = hi[0];                                                         // what moves up a word: < 2^18
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        const u32 x = lo[q] + c;
        c = hi[q] + (x < c ? 1u : 0u);
        v[q] = x;
    }
    v[8] = c;
    const u32 m = v[0] * mf_rinv32();
    u64 t = (u64)m * mf_r32(0) + v[0];                                     // low word zero
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        t = (u64)m * mf_r32(q) + ((u64)v[q] + (t >> 32));                  // <= (2^32 - 1)^2 + 2 (2^32 - 1)
        out[q - 1] = (u32)t;
    }
    out[7] = v[8] + (u32)(t >> 32);
}
__device__ __forceinline__ fe mf_words_to_fe(const u32 (&w)[8]) {           // value < 2^256 -> limbs, normalised
    fe x;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int bit = LB * k, wi = bit >> 5, s = bit & 31;
        u32 v = w[wi] >> s;
        if (s > 32 - LB && wi + 1 < 8) v |= w[wi + 1] << (32 - s);
        x.l[k] = v & LMASK;
    }
    return x;
}
__device__ __forceinline__ void mf_b_from_words(u32 (&w)[8], mf_v4i& b0, mf_v4i& b1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] ^= 0x80808080u;
#pragma unroll
    for (int i = 0; i < 4; ++i) mf_swap32(w[i], w[4 + i]);
    b0 = mf_v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    b1 = mf_v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
}
__device__ __forceinline__ fe mf_fe_from_b(const mf_v4i& b0, const mf_v4i& b1) {   // the lane's own word back out of its two operands
    u32 w[8] = {(u32)b0[0], (u32)b0[1], (u32)b0[2], (u32)b0[3], (u32)b1[0], (u32)b1[1], (u32)b1[2], (u32)b1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) mf_swap32(w[i], w[4 + i]);                 // the swap is its own inverse
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] ^= 0x80808080u;
    return mf_words_to_fe(w);
}
// two tiles' accumulators of one output -> the lane's own eight words of it
__device__ __forceinline__ void mf_output_words(const mf_v16i& acc0, const mf_v16i& acc1, const u64* __restrict__ Ko, int one, int s16, u32 (&out)[8]) {
    u64 Wa[4], Wb[4];
    mf_words(acc0, Ko, one, s16, Wa);
    mf_words(acc1, Ko, one, s16, Wb);
    u32 lo[8], hi[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        lo[g] = (u32)Wa[g]; hi[g] = (u32)(Wa[g] >> 32); lo[4 + g] = (u32)Wb[g]; hi[4 + g] = (u32)(Wb[g] >> 32);
        mf_swap32(lo[g], lo[4 + g]); mf_swap32(hi[g], hi[4 + g]);         // now [0..3] = the lane's own words 0..3, [4..7] = its own words 4..7, in both halves
    }
    mf_reduce_words(lo, hi, out);
}
// All n_rp sparse rounds on st (words < 2^256 in; < 2^242 + r out, limbs normalised).  tab: the rounds' blocks (mf_build_sparse, global
// memory), ident: the unit's fragment, abuf: two blocks' worth of LDS.  The whole block of 256 threads must be here.
template <int T>
__device__ __forceinline__ void mf_sparse(fe (&st)[T], const mf_v4i* __restrict__ tab, const mf_v4i* __restrict__ ident, mf_v4i* abuf, u32 n_rp) {
    constexpr int PER = (2 * T - 1) * 64 + 4 * T, NLD = (PER + 255) / 256;   // a round's block in 16-byte words
    const int lane = threadIdx.x & 63, H = lane >> 5;
    mf_v4i pf[NLD];
    auto fetch = [&](u32 r) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int e = threadIdx.x + 256 * k; if (NLD * 256 == PER || e < PER) pf[k] = tab[(size_t)r * PER + e]; }
    };
    auto stash = [&](u32 r) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int e = threadIdx.x + 256 * k; if (NLD * 256 == PER || e < PER) abuf[(r & 1) * PER + e] = pf[k]; }
    };
    fetch(0);
    const mf_v4i aI = ident[lane];
    mf_v4i B0[T], B1[T];                                                   // [0] unused: word 0 goes through the S-box in limbs
    fh_static_for<1, T>([&](auto J) { constexpr int j = decltype(J)::value; mf_make_b(st[j], B0[j], B1[j]); });
    fe s0 = st[0];
    stash(0);
    __syncthreads();
    int one = 1, s16 = 65536;
    asm volatile("" : "+v"(one), "+v"(s16));
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (u32 r = 0; r < n_rp; ++r) {
        if (r + 1 < n_rp) fetch(r + 1);
        const mf_v4i* __restrict__ fr = abuf + (r & 1) * PER + lane;
        const u64* __restrict__ Kr = reinterpret_cast<const u64*>(abuf + (r & 1) * PER + (2 * T - 1) * 64) + 4 * H;
        // the row's products with the words that do not pass the S-box go first: the matrix pipe works through them under the S-box
        mf_v16i row0 = {}, row1 = {};
        fh_static_for<1, T>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const mf_v4i a = fr[j * 64];
            row0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[j], row0, 0, 0, 0);
            row1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[j], row1, 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        pow5(s0);
        mf_v4i P0, P1;
        mf_make_b(s0, P0, P1);
        {
            const mf_v4i a0 = fr[0];
            row0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, P0, row0, 0, 0, 0);
            row1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, P1, row1, 0, 0, 0);
        }
        // columns k: S'[k] x_0^5 + x_k, the products of column k + 1 issued before column k is put together
        mf_v16i ca0[2], ca1[2];
        auto issue = [&](auto KK, int slot) {
            constexpr int k = decltype(KK)::value;
            const mf_v4i a = fr[(T + k - 1) * 64];
            ca0[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, P0, (mf_v16i{}), 0, 0, 0);
            ca1[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, P1, (mf_v16i{}), 0, 0, 0);
            ca0[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aI, B0[k], ca0[slot], 0, 0, 0);
            ca1[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aI, B1[k], ca1[slot], 0, 0, 0);
        };
        issue(std::integral_constant<int, 1>{}, 1);
        __builtin_amdgcn_sched_barrier(0);
        {
            u32 w[8];
            mf_output_words(row0, row1, Kr, one, s16, w);
            s0 = mf_words_to_fe(w);
        }
        fh_static_for<1, T>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            if constexpr (k + 1 < T) { issue(std::integral_constant<int, k + 1>{}, (k + 1) & 1); __builtin_amdgcn_sched_barrier(0); }
            u32 w[8];
            mf_output_words(ca0[k & 1], ca1[k & 1], Kr + 8 * k, one, s16, w);
            mf_b_from_words(w, B0[k], B1[k]);
        });
        if (r + 1 < n_rp) stash(r + 1);
        __syncthreads();
    }
    st[0] = s0;
    fh_static_for<1, T>([&](auto J) { constexpr int j = decltype(J)::value; st[j] = mf_fe_from_b(B0[j], B1[j]); });
}
