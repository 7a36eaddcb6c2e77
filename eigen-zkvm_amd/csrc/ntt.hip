// Goldilocks NTT / iNTT / LDE for gfx950 (MI355X).
//
// Replaces starky/src/fft_p.rs:174-355 (_fft, fft, ifft, interpolate) and fft_worker.rs:25-72.
// Same contract: row-major [1<<nbits][n_pols] matrix of canonical u64, natural order in and out,
// forward root w = MG.0[nbits] (constant.rs:54-68), inverse = 1/N * NTT with w^-1
// (== the reference's (n-BR(i))%n input permutation, fft_p.rs:124-142), LDE on the coset
// 49*<w_ext> (fft_p.rs:255-355).
//
// Design (not the reference's bit-reverse + blocked butterflies + transposes):
//   * Stockham auto-sort decomposition N = R_1*R_2*...; every pass reads R rows of `inner`
//     contiguous words (row stride N/R * n_pols) and writes runs of >= s*n_pols contiguous words,
//     so there is no bit-reversal pass and no transpose pass; natural order falls out.
//       pass with stride s (product of earlier radices), L = N/s, p < L/R, q < s:
//         out[(p*R + k)*s + q] = w_L^(p*k) * sum_j in[j*N/R + p*s + q] * w_R^(j*k)
//   * one workgroup = one tile of R (<=256) points x C = 4096/R adjacent lanes (a lane is one
//     (row-within-stride, column) pair, i.e. one contiguous u64 of the matrix row range);
//     each thread owns 16 points: radix-2^LOGA DIF in registers -> twiddle -> ONE LDS
//     transpose -> radix-2^LOGB DIF in registers -> inter-pass twiddle -> store.
//     A 2^24 transform is three such passes (8+8+8 bits), 2 LDS accesses / element / pass.
//   * twiddles w_N^e come from a two-level table (4096 + N/4096 entries, L2 resident); the
//     per-thread chain over the 16 outputs needs 2 lookups + 1 multiply per element.
//   * HBM-bound: 16 B / element / pass; bytes per transform = 16 * passes * N * n_pols.
#include "zk_internal.h"
#include "ntt_reg.hip.h"
#include <map>
#include <mutex>
#include <vector>

namespace zk {

namespace {

// elements per tile (a workgroup of NTT_TILE / 16 lanes, 8 bytes of LDS per element); tiles of 2048 and 1024 measured 24 % / 77 % slower
// (profiles/r05/ntt_bound.md; the switches of those runs: tools/experiments/rejected_switches_r02_r05.patch)
constexpr int NTT_TILE = 4096;
constexpr int TW_LO_BITS = 12;
constexpr int TW_LO = 1 << TW_LO_BITS;

struct PassParams {
    const u64* in;
    u64* out;
    const u64* w256;   // w_256^e, e < 256 (direction-specific)
    const u64* tw_lo;  // w_N^i, i < 4096
    const u64* tw_hi;  // w_N^(4096 i)
    const u64* tw_mid; // optional: w_N^(i << dshift), i < 2^16 -- direct inter-pass twiddles when L <= 2^16
    u64 pre_scale;     // constant folded into sub-step A (w256 is then the pre-scaled table); 1 = none
    u32 dshift;
    const u64* sc_lo;  // optional output scaling c * g^k (k = output row): g^i * c, i < 4096
    const u64* sc_hi;  // g^(4096 i)
    u64 sc_step;       // g^(RA * s)
    u64 out_scale;     // plain constant scaling (1 = none), used when sc_lo == nullptr
    u64 inner;         // (N/R) * n_pols : contiguous words per transform-axis index j
    u64 valid_in;      // words of `in` that exist; beyond that the input is implicit zero (LDE)
    u64 s_np;          // s * n_pols
    u32 np;
    u32 log_s;
    u32 has_tw;        // L > R (not the last pass)
};

__device__ __forceinline__ u64 tab2(const u64* __restrict__ lo, const u64* __restrict__ hi, u64 e) {
    return gl::mul(hi[e >> TW_LO_BITS], lo[e & (TW_LO - 1)]);
}

template <int LOGA, int LOGB, bool KMODE, bool INV>
__global__ __launch_bounds__(NTT_TILE / 16) void ntt_pass_kernel(const PassParams P) {
    constexpr int LOGR = LOGA + LOGB, R = 1 << LOGR, RA = 1 << LOGA, RB = 1 << LOGB;
    constexpr int C = NTT_TILE / R;         // lanes per tile
    constexpr int GA = 16 / RA, GB = 16 / RB;  // independent sub-transforms per thread
    constexpr int TG = R / 16;                 // threads along the transform axis
    constexpr int PAD = KMODE ? 512 / R : 0;   // words; spreads kappa_a rows over banks (KMODE reads)
    constexpr int ROW = RB * C + PAD;
    static_assert(LOGR >= 4 && LOGR <= 8 && LOGA <= 4 && LOGB <= 4, "tile shape");
    __shared__ u64 lds[RA * ROW];

    const int t = threadIdx.x;
    const u64 u0 = (u64)blockIdx.x * C;

    {   // ---- sub-step A: RA-point transforms over ja (j = ja*RB + jb), twiddle w_R^(jb*ka)
        const int c = t % C, ta = t / C;
        const u64 u = u0 + c;
        const bool live = u < P.inner;
        u64 x[GA][RA];
        // Whole tile inside the matrix and inside the rows that exist (wave-uniform test): sixteen unconditional loads,
        // issued back to back.  Otherwise (ragged last tile, or the zero-padded half of an extension's first forward
        // pass) every load still goes out unconditionally -- of word 0 where there is nothing to read -- and is zeroed afterwards: a
        // branch per element would serialise the loads behind each other's latency.
        const bool full = u0 + C <= P.inner && (u64)(R - 1) * P.inner + u0 + C <= P.valid_in;
        const u64* __restrict__ base = P.in + (u64)(ta * GA) * P.inner + (live ? u : 0);
        if (full) {
#pragma unroll
            for (int g = 0; g < GA; ++g)
#pragma unroll
                for (int ja = 0; ja < RA; ++ja) x[g][ja] = base[(u64)(ja * RB + g) * P.inner];
        } else {
#pragma unroll
            for (int g = 0; g < GA; ++g)
#pragma unroll
                for (int ja = 0; ja < RA; ++ja) {
                    const u64 off = (u64)(ja * RB + g) * P.inner;
                    const bool ok = live && (u64)(ta * GA) * P.inner + off + u < P.valid_in;
                    const u64 v = *(ok ? base + off : P.in);       // word 0 always exists
                    x[g][ja] = ok ? v : 0;
                }
        }
#pragma unroll
        for (int g = 0; g < GA; ++g) ntt_reg<LOGA, INV>(x[g]);
        // Passes of <= 6 bits: w_R is a power of two (w_64 = 2^39, ntt_reg.hip.h), so the twiddle w_R^(jb ka) between the two halves is a
        // shift like the butterflies' own -- once jb is a compile-time number.  jb = ta GA + g and ta = t / C with C = 4096 / R >= 64
        // lanes: the same in every lane of a wave, so a switch over ta (R / 16 <= 4 cases) costs no divergence and turns the 15 general
        // products of a lane into 15 shifts.  (Not for the pre-scaled last pass of a plain inverse transform: its 1/N rides on the table.)
        if (LOGR <= 6 && C >= 64 && LOGB > 0 && P.pre_scale == 1) {
            static_for<0, TG>([&](auto TA) {
                constexpr int ta_c = decltype(TA)::value;
                if (ta == ta_c) {
                    static_for<0, GA>([&](auto GI) {
                        constexpr int g = decltype(GI)::value, jb = ta_c * GA + g;
                        static_for<0, RA>([&](auto KA) {
                            constexpr int ka = decltype(KA)::value;
                            constexpr int e0 = (39 * (64 / (R > 64 ? 64 : R)) * jb * ka) % 192, e = INV ? (192 - e0) % 192 : e0;   // w_R^(+-jb ka) = 2^e
                            u64 v = x[g][bitrev_c(ka, LOGA)];
                            if constexpr (e >= 96) v = gl::neg(gl::mul_pow2<e - 96>(v));
                            else v = gl::mul_pow2<e>(v);
                            lds[ka * ROW + jb * C + c] = v;
                        });
                    });
                }
            });
        } else
        {
#pragma unroll
            for (int g = 0; g < GA; ++g) {
                const int jb = ta * GA + g;
#pragma unroll
                for (int ka = 0; ka < RA; ++ka) {
                    u64 v = x[g][bitrev_c(ka, LOGA)];
                    if (LOGB > 0 && ka > 0) v = gl::mul(v, P.w256[(jb * ka) << (8 - LOGR)]);
                    else if (P.pre_scale != 1) v = gl::mul(v, P.pre_scale);
                    lds[ka * ROW + jb * C + c] = v;
                }
            }
        }
    }
    __syncthreads();
    {   // ---- sub-step B: RB-point transforms over jb -> kappa = RA*kb + ka
        int c, tb;
        if (KMODE) { tb = t % TG; c = t / TG; } else { c = t % C; tb = t / C; }
        const u64 u = u0 + c;
        u64 y[GB][RB];
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int ka = tb + g * TG;
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) y[g][jb] = lds[ka * ROW + jb * C + c];
        }
        const u64 uc = u < P.inner ? u : 0;
        // p = uc / (s n_pols): a shift for one column; a 32-bit division when the matrix has fewer than 2^32 words per transform-axis
        // index (every size in use); the 64-bit division otherwise -- uniform branches, the general case cost ~100 instructions per lane
        u64 p, rem;
        if (P.np == 1) { p = uc >> P.log_s; rem = uc & (((u64)1 << P.log_s) - 1); }
        else if ((P.inner >> 32) == 0) { const u32 p32 = (u32)uc / (u32)P.s_np; p = p32; rem = (u32)uc - p32 * (u32)P.s_np; }
        else
        { p = uc / P.s_np; rem = uc - p * P.s_np; }
        u64 tw[GB][RB];
        if (P.tw_mid) {  // L <= 2^16: every twiddle w_L^(p*kappa) is one load from the 512 KB table (L2), no chain;
                         // issued here so that the radix-2^LOGB butterflies below hide the latency
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int ka = tb + g * TG;
#pragma unroll
                for (int kb = 0; kb < RB; ++kb) tw[g][kb] = P.tw_mid[((p * (u64)(RA * kb + ka)) << P.log_s) >> P.dshift];
            }
        }
        // (a uniform run-time branch on purpose: as `if constexpr` the compiler sinks these loads below the
        // butterflies to save registers and the L2 latency lands on the critical path again)
#pragma unroll
        for (int g = 0; g < GB; ++g) ntt_reg<LOGB, INV>(y[g]);
        if (u >= P.inner) return;

        u64* __restrict__ outp = P.out + p * R * P.s_np + rem;
        const u64 kstride = P.s_np;                    // words between the outputs kappa and kappa + 1 of a lane
        const u64 row_q = !P.sc_lo ? 0 : P.np == 1 ? rem : (P.inner >> 32) == 0 ? (u64)((u32)rem / P.np) : rem / P.np;  // output row = kappa*s + row_q (last pass: p == 0)
        if (P.tw_mid) {
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int ka = tb + g * TG;
#pragma unroll
                for (int kb = 0; kb < RB; ++kb) {
                    u64 v = y[g][bitrev_c(kb, LOGB)];
                    v = gl::mul(v, tw[g][kb]);
                    outp[(u64)(RA * kb + ka) * kstride] = v;
                }
            }
            return;
        }
        u64 tw_step = 1;
        if (P.has_tw) tw_step = tab2(P.tw_lo, P.tw_hi, (p * RA) << P.log_s);
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int ka = tb + g * TG;
            u64 f = P.out_scale;
            if (P.has_tw) f = tab2(P.tw_lo, P.tw_hi, (p * ka) << P.log_s);
            if (P.sc_lo) f = tab2(P.sc_lo, P.sc_hi, ((u64)ka << P.log_s) + row_q);
            const bool scaled = P.has_tw || P.sc_lo || P.out_scale != 1;
            const u64 fstep = P.sc_lo ? P.sc_step : tw_step;
#pragma unroll
            for (int kb = 0; kb < RB; ++kb) {
                u64 v = y[g][bitrev_c(kb, LOGB)];
                if (scaled) {
                    v = gl::mul(v, f);
                    if (kb + 1 < RB) f = gl::mul(f, fstep);
                }
                outp[(u64)(RA * kb + ka) * kstride] = v;
            }
        }
    }
}

// Direct evaluation for transforms shorter than one register tile (nbits < 4): out[k][c] =
// scale(k) * sum_{i < n_in} in[i][c] * w^(i k).  A handful of rows; never on a hot path.
__global__ void ntt_small_kernel(const u64* in, u64* out, u32 np, u32 n_in, u32 n, u64 w, u64 g, u64 cst) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n * np) return;
    const u32 k = tid / np, c = tid % np;
    const u64 wk = gl::pow(w, k);
    u64 acc = 0, cur = 1;
    for (u32 i = 0; i < n_in; ++i) { acc = gl::add(acc, gl::mul(in[(u64)i * np + c], cur)); cur = gl::mul(cur, wk); }
    out[(u64)k * np + c] = gl::mul(acc, gl::mul(cst, gl::pow(g, k)));
}

// ---- host side: tables + plan -------------------------------------------------------------
struct Tables { u64 *w256 = nullptr, *w256s = nullptr, *lo = nullptr, *hi = nullptr, *mid = nullptr; u32 dshift = 0; };
struct ScaleTables { u64 *lo = nullptr, *hi = nullptr; };

std::mutex g_mu;
std::map<std::pair<int, std::pair<u32, int>>, Tables> g_tables;      // (device,(nbits,inverse))
std::map<std::pair<int, std::pair<u32, u64>>, ScaleTables> g_scales;  // (device,(nbits,g))

u64* upload(const std::vector<u64>& v) {
    u64* d = nullptr;
    ZK_HIP(hipMalloc((void**)&d, v.size() * sizeof(u64)));
    ZK_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(u64), hipMemcpyHostToDevice));
    return d;
}

void two_level(u64 base, u64 cst, u32 nbits, std::vector<u64>& lo, std::vector<u64>& hi) {
    lo.resize(TW_LO);
    u64 c = cst;
    for (int i = 0; i < TW_LO; ++i) { lo[i] = c; c = gl::hmul(c, base); }
    size_t nh = nbits > (u32)TW_LO_BITS ? (size_t)1 << (nbits - TW_LO_BITS) : 1;
    hi.resize(nh);
    u64 step = gl::hpow(base, TW_LO);
    c = 1;
    for (size_t i = 0; i < nh; ++i) { hi[i] = c; c = gl::hmul(c, step); }
}

Tables get_tables(u32 nbits, bool inverse) {
    int dev; ZK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, std::make_pair(nbits, (int)inverse));
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return it->second;
    u64 w = gl::hroot(nbits), w8 = gl::hroot(8);
    if (inverse) { w = gl::hinv(w); w8 = gl::hinv(w8); }
    std::vector<u64> t256(256), lo, hi;
    u64 c = 1;
    for (int i = 0; i < 256; ++i) { t256[i] = c; c = gl::hmul(c, w8); }
    two_level(w, 1, nbits, lo, hi);
    Tables T; T.w256 = upload(t256); T.lo = upload(lo); T.hi = upload(hi);
    if (inverse) {  // w_256^-e / N: folds the inverse transform's 1/N into sub-step A of its last pass
        const u64 ninv = gl::hinv((1ull << nbits) % GL_P);
        std::vector<u64> t256s(256);
        for (int i = 0; i < 256; ++i) t256s[i] = gl::hmul(t256[i], ninv);
        T.w256s = upload(t256s);
    }
    {   // direct table for passes with L <= 2^16: w_N^(i << dshift)
        T.dshift = nbits > 16 ? nbits - 16 : 0;
        const size_t nm = (size_t)1 << (nbits - T.dshift);
        std::vector<u64> mid(nm);
        const u64 step = gl::hpow(w, 1ull << T.dshift);
        c = 1;
        for (size_t i = 0; i < nm; ++i) { mid[i] = c; c = gl::hmul(c, step); }
        T.mid = upload(mid);
    }
    g_tables[key] = T;
    return T;
}

ScaleTables get_scale(u32 nbits, u64 g, u64 cst) {  // cst * g^k, k < 2^nbits
    int dev; ZK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, std::make_pair(nbits, g));
    auto it = g_scales.find(key);
    if (it != g_scales.end()) return it->second;
    std::vector<u64> lo, hi;
    two_level(g, cst, nbits, lo, hi);
    ScaleTables S; S.lo = upload(lo); S.hi = upload(hi);
    g_scales[key] = S;
    return S;
}

template <int LOGA, int LOGB>
void launch_pass(const PassParams& P, bool kmode, bool inverse, hipStream_t st) {
    constexpr int C = NTT_TILE >> (LOGA + LOGB);
    const u64 blocks = (P.inner + C - 1) / C;
    ZK_REQUIRE(blocks < (1ull << 31), "ntt: grid too large");
    const dim3 g((u32)blocks), b(NTT_TILE / 16);
    const int variant = (kmode ? 2 : 0) | (inverse ? 1 : 0);
#define ZK_PASS(K, I) hipLaunchKernelGGL((ntt_pass_kernel<LOGA, LOGB, K, I>), g, b, 0, st, P)
    switch (variant) {
        case 0: ZK_PASS(false, false); break; case 1: ZK_PASS(false, true); break;
        case 2: ZK_PASS(true, false); break;  default: ZK_PASS(true, true); break;
    }
#undef ZK_PASS
    ZK_HIP(hipGetLastError());
}

void launch_pass_logr(int logr, const PassParams& P, bool kmode, bool inverse, hipStream_t st) {
    switch (logr) {
        case 4: launch_pass<2, 2>(P, kmode, inverse, st); break;
        case 5: launch_pass<3, 2>(P, kmode, inverse, st); break;
        case 6: launch_pass<3, 3>(P, kmode, inverse, st); break;
        case 7: launch_pass<4, 3>(P, kmode, inverse, st); break;
        case 8: launch_pass<4, 4>(P, kmode, inverse, st); break;
        default: throw Error("ntt: unsupported pass radix");
    }
}

// radices of the passes of a 2^nbits transform (each 4..8 bits); empty for nbits < 4
std::vector<int> plan(u32 nbits) {
    std::vector<int> r;
    if (nbits < 4) return r;
    int np = (nbits + 7) / 8;
    int base = nbits / np, extra = nbits % np;
    for (int i = 0; i < np; ++i) r.push_back(base + (i < extra ? 1 : 0));
    return r;
}

// One transform of size 2^nbits whose input holds `valid_rows` rows (rest implicit zero).
// bufs: sequence of distinct buffers the passes ping-pong through, bufs[0] = input,
// bufs.back() = output.  scale: optional (g, cst) output scaling cst*g^row; out_scale: constant.
struct Scale { bool on = false; u64 g = 1, cst = 1; };

void run_transform(const u64* in, u64* a, u64* b, /* ping-pong, result must land in `a` */
                   u32 n_pols, u32 nbits, u64 valid_rows, bool inverse, Scale sc, u64 out_scale, hipStream_t st) {
    const u64 n = 1ull << nbits;
    if (nbits < 4) {
        u64 w = gl::hroot(nbits);
        if (inverse) w = gl::hinv(w);
        u64 cst = sc.on ? sc.cst : out_scale;
        u64 g = sc.on ? sc.g : 1;
        u32 total = (u32)(n * n_pols);
        hipLaunchKernelGGL(ntt_small_kernel, dim3((total + 255) / 256), dim3(256), 0, st, in, a, n_pols,
                           (u32)valid_rows, (u32)n, w, g, cst);
        ZK_HIP(hipGetLastError());
        return;
    }
    const std::vector<int> radices = plan(nbits);
    const int np = (int)radices.size();
    Tables T = get_tables(nbits, inverse);
    ScaleTables S;
    if (sc.on) S = get_scale(nbits, sc.g, sc.cst);
    u32 log_s = 0;
    const u64* cur = in;
    for (int i = 0; i < np; ++i) {
        const int logr = radices[i];
        u64* dstbuf = ((np - 1 - i) % 2 == 0) ? a : b;
        const bool last = (i == np - 1);
        PassParams P{};
        P.in = cur; P.out = dstbuf;
        P.w256 = T.w256; P.tw_lo = T.lo; P.tw_hi = T.hi;
        P.sc_lo = (last && sc.on) ? S.lo : nullptr;
        P.sc_hi = (last && sc.on) ? S.hi : nullptr;
        const int loga = (logr + 1) / 2;
        P.sc_step = (last && sc.on) ? gl::hpow(sc.g, (1ull << loga) << log_s) : 1;
        P.out_scale = 1; P.pre_scale = 1;
        if (last && !sc.on && out_scale != 1) {
            // the only constant scaling in use is the inverse transform's 1/N (ntt_dev)
            ZK_REQUIRE(inverse && out_scale == gl::hinv(n % GL_P), "ntt: unsupported constant scaling");
            P.w256 = T.w256s; P.pre_scale = out_scale;
        }
        P.inner = (n >> logr) * n_pols;
        P.valid_in = (i == 0) ? valid_rows * n_pols : n * n_pols;
        P.s_np = ((u64)1 << log_s) * n_pols;
        P.np = n_pols;
        P.log_s = log_s;
        P.has_tw = last ? 0 : 1;
        P.dshift = T.dshift;
        P.tw_mid = (!last && log_s >= T.dshift) ? T.mid : nullptr;  // L = N >> log_s <= 2^16
        const bool kmode = P.s_np < 16;
        launch_pass_logr(logr, P, kmode, inverse, st);
        cur = dstbuf;
        log_s += logr;
    }
}

}  // namespace

int ntt_num_passes(uint32_t nbits) { return nbits < 4 ? 1 : (int)plan(nbits).size(); }

const u64* ntt_w256_table(bool inverse) { return get_tables(8, inverse).w256; }

void ntt_dev(const u64* d_src, u64* d_dst, u64* d_tmp, uint32_t n_pols, uint32_t nbits, bool inverse, hipStream_t st) {
    ZK_REQUIRE(nbits <= 32, "ntt: nbits > 32");
    if (n_pols == 0) return;
    ZK_REQUIRE(d_src != d_dst, "ntt: dst may not alias src");
    u64 out_scale = inverse ? gl::hinv((1ull << nbits) % GL_P) : 1;
    run_transform(d_src, d_dst, d_tmp, n_pols, nbits, 1ull << nbits, inverse, Scale{}, out_scale, st);
}

void lde_dev(const u64* d_src, u64* d_dst, u64* d_tmp, uint32_t n_pols, uint32_t nbits, uint32_t nbits_ext, hipStream_t st) {
    ZK_REQUIRE(nbits_ext <= 32 && nbits <= nbits_ext, "lde: need nbits <= nbits_ext <= 32");
    if (n_pols == 0) return;  // fft_p.rs:262-264
    ZK_REQUIRE(d_src != d_dst, "lde: dst may not alias src");
    const u64 n = 1ull << nbits;
    // (Blow-up 2 as two size-N coset transforms with interleaved output rows instead of one 2N-point transform over a half-zero input was
    // measured in round 3: bit-exact, 18.7 against 15.2 ms for 19 columns at 2^24 -- tools/experiments/rejected_switches_r02_r05.patch.)
    // coefficients * 49^i / N  (fft_p.rs:144-172): the inverse transform's last pass applies it.
    Scale sc; sc.on = true; sc.g = 49; sc.cst = gl::hinv(n % GL_P);
    const int fwd_passes = ntt_num_passes(nbits_ext);
    // forward transform must end in d_dst; its input (the coefficient buffer) must differ from
    // the first forward pass's output: forward pass i writes (fwd_passes-1-i)%2==0 ? dst : tmp.
    u64* first_fwd_out = ((fwd_passes - 1) % 2 == 0) ? d_dst : d_tmp;
    u64* coef = (first_fwd_out == d_dst) ? d_tmp : d_dst;
    u64* other = (coef == d_dst) ? d_tmp : d_dst;
    run_transform(d_src, coef, other, n_pols, nbits, n, true, sc, 1, st);
    run_transform(coef, d_dst, d_tmp, n_pols, nbits_ext, n, false, Scale{}, 1, st);
}

}  // namespace zk
