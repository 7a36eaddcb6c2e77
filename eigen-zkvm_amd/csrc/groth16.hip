// Groth16 around the multi-scalar sums on gfx950 -- SURVEY.md 8(f)-2: `zkit groth16_prove`
// (groth16/src/api.rs:144-205 -> groth16.rs:88-96 -> bellman_ce::groth16::create_random_proof, third-party) with
// everything after witness generation resident on the device:
//   file formats     .r1cs (algebraic/src/r1cs_file.rs:50-118, 185-270), .wtns (algebraic/src/reader.rs:86-137),
//                    bellman's Parameters (groth16/src/api.rs:545-550; pairing_ce's uncompressed points, the
//                    encoding of groth16/test-vectors/verification_key*.bin)
//   circuit          algebraic/src/circom_circuit.rs:94-160 + the `input * 0 = 0` rows bellman's prover appends
//   row evaluations  one lane per row of the three CSR matrices                 (frntt_impl.hip.h)
//   quotient         7 transforms over Fr + the pointwise step                  (frntt_impl.hip.h)
//   h, l, a, b_g1, b_g2 sums and the final assembly through msm.hip
//   proof.json       groth16/src/json_utils.rs:305-315
// r and s are taken from the caller (the reference draws them from OsRng, api.rs:172); everything else is a
// function of (key, circuit, witness).
#include "zk_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace zk {
namespace g16 {

namespace {
struct Reader {
    const uint8_t* p; size_t n, o = 0; const char* what;
    void need(size_t k) const { if (k > n - o) throw std::runtime_error(std::string(what) + ": truncated file"); }
    uint32_t u32le() { need(4); uint32_t v; std::memcpy(&v, p + o, 4); o += 4; return v; }
    uint64_t u64le() { need(8); uint64_t v; std::memcpy(&v, p + o, 8); o += 8; return v; }
    uint32_t u32be() { need(4); uint32_t v = ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | p[o + 3]; o += 4; return v; }
    const uint8_t* take(size_t k) { need(k); const uint8_t* q = p + o; o += k; return q; }
};
bool words_lt(const u32* a, const u32* b, int n) {
    for (int i = n - 1; i >= 0; --i) { if (a[i] < b[i]) return true; if (a[i] > b[i]) return false; }
    return false;
}
}  // namespace

std::string words_to_dec(const u32* w, int n) {
    std::vector<u32> v(w, w + n);
    std::string out;
    while (true) {
        bool zero = true;
        uint64_t rem = 0;
        for (int i = n - 1; i >= 0; --i) {
            const uint64_t cur = (rem << 32) | v[i];
            v[i] = (u32)(cur / 1000000000u); rem = cur % 1000000000u;
            zero = zero && v[i] == 0;
        }
        char buf[16];
        if (zero) { snprintf(buf, sizeof buf, "%u", (unsigned)rem); out = buf + out; break; }
        snprintf(buf, sizeof buf, "%09u", (unsigned)rem); out = buf + out;
    }
    return out;
}

// r1cs_file.rs:185-270 from_reader (sections may come in any order; custom gates are not supported there either)
static R1cs parse_r1cs(const uint8_t* b, size_t len, const std::vector<u32>& modulus) {
    Reader rd{b, len, 0, "r1cs"};
    if (std::memcmp(rd.take(4), "r1cs", 4) != 0) throw std::runtime_error("r1cs: Invalid magic number");
    if (rd.u32le() != 1) throw std::runtime_error("r1cs: Unsupported version");
    const uint32_t n_sec = rd.u32le();
    std::map<uint32_t, std::pair<size_t, uint64_t>> sec;
    for (uint32_t i = 0; i < n_sec; ++i) {
        const uint32_t t = rd.u32le(); const uint64_t sz = rd.u64le();
        sec[t] = {rd.o, sz};
        rd.take(sz);
    }
    if (!sec.count(1) || !sec.count(2)) throw std::runtime_error("r1cs: header or constraint section missing");
    Reader h{b + sec[1].first, (size_t)sec[1].second, 0, "r1cs header"};
    const uint32_t fs = h.u32le();
    if (sec[1].second != 32 + (uint64_t)fs) throw std::runtime_error("r1cs: Invalid header section size");
    if (fs != 32) throw std::runtime_error("r1cs: field size " + std::to_string(fs) + " is not 32 bytes");
    const uint8_t* prime = h.take(fs);
    if (std::memcmp(prime, modulus.data(), 32) != 0) throw std::runtime_error("r1cs: the file's prime is not the scalar field of the selected curve");
    R1cs rc;
    rc.n_wires = h.u32le(); rc.n_pub_out = h.u32le(); rc.n_pub_in = h.u32le(); rc.n_prv_in = h.u32le();
    (void)h.u64le();
    const uint32_t n_cons = h.u32le();
    Reader c{b + sec[2].first, (size_t)sec[2].second, 0, "r1cs constraints"};
    rc.rows.resize(n_cons);
    for (uint32_t i = 0; i < n_cons; ++i)
        for (int w = 0; w < 3; ++w) {
            const uint32_t nv = c.u32le();
            std::vector<std::pair<uint32_t, const uint8_t*>> terms(nv);
            for (uint32_t k = 0; k < nv; ++k) { terms[k].first = c.u32le(); terms[k].second = c.take(32); }
            std::stable_sort(terms.begin(), terms.end(), [](const auto& x, const auto& y) { return x.first < y.first; });   // r1cs_file.rs:83
            Lc& lc = rc.rows[i].lc[w];
            for (auto& t : terms) {
                u32 v[8]; std::memcpy(v, t.second, 32);
                if (!words_lt(v, modulus.data(), 8)) throw std::runtime_error("r1cs: coefficient is not a canonical field element");   // Fr::from_repr
                lc.col.push_back(t.first); lc.coeff.insert(lc.coeff.end(), v, v + 8);
            }
        }
    return rc;
}

// pairing_ce's uncompressed encoding: big-endian canonical coordinates (G2: x.c1, x.c0, y.c1, y.c0), bit 6 of the
// first byte = infinity, bit 7 = compressed (rejected)
static void parse_points(Reader& rd, uint64_t count, int coord_bytes, bool g2, PointVec& out) {
    const int nc = g2 ? 4 : 2, cw = coord_bytes / 4;
    out.n = count; out.w.assign(count * nc * cw, 0); out.inf.assign(count, 0);
    for (uint64_t i = 0; i < count; ++i) {
        const uint8_t* p = rd.take((size_t)nc * coord_bytes);
        if (p[0] & 0x80) throw std::runtime_error("proving key: compressed point where an uncompressed one is expected");
        if (p[0] & 0x40) { out.inf[i] = 1; continue; }
        for (int c = 0; c < nc; ++c) {
            const int dst = g2 ? (c ^ 1) : c;                       // swap c1/c0 within x and within y
            const uint8_t* q = p + (size_t)c * coord_bytes;
            u32* w = out.w.data() + (i * nc + dst) * cw;
            for (int k = 0; k < cw; ++k) {
                const uint8_t* e = q + coord_bytes - 4 * (k + 1);
                w[k] = ((u32)e[0] << 24) | ((u32)e[1] << 16) | ((u32)e[2] << 8) | e[3];
            }
        }
    }
}
// bellman groth16 Parameters::read (vk, then h, l, a, b_g1, b_g2, each with a big-endian u32 count)
static Params parse_params(const uint8_t* b, size_t len, int coord_bytes) {
    Reader rd{b, len, 0, "proving key"};
    Params P;
    const bool g2[6] = {false, false, true, true, false, true};
    for (int i = 0; i < 6; ++i) parse_points(rd, 1, coord_bytes, g2[i], P.vk[i]);
    parse_points(rd, rd.u32be(), coord_bytes, false, P.ic);
    parse_points(rd, rd.u32be(), coord_bytes, false, P.h);
    parse_points(rd, rd.u32be(), coord_bytes, false, P.l);
    parse_points(rd, rd.u32be(), coord_bytes, false, P.a);
    parse_points(rd, rd.u32be(), coord_bytes, false, P.b_g1);
    parse_points(rd, rd.u32be(), coord_bytes, true, P.b_g2);
    if (rd.o != len) throw std::runtime_error("proving key: trailing bytes");
    return P;
}
}  // namespace g16

namespace bn254fr {
#define ZK_FR29_FIELD 254
#include "fr29_consts.hip.h"
#include "fe29_impl.hip.h"
#define FRN_S 28
#define FRN_ROOT 0xb639feb8u, 0x9632c7c5u, 0x0d0ff299u, 0x985ce340u, 0x01b0ecd8u, 0xb2dd8800u, 0x6d98ce29u, 0x1d69070du   // 7^((r-1)/2^28) * 2^256
#define FRN_FN(name) name
#include "frntt_impl.hip.h"
#define G16_CW 8
#define G16_MSM_G1 msm_g1_bn254_dev
#define G16_MSM_G2 msm_g2_bn254_dev
#define G16_MSM_G1_TABLE_BYTES msm_g1_bn254_fixed_table_bytes
#define G16_MSM_G2_TABLE_BYTES msm_g2_bn254_fixed_table_bytes
#define G16_MSM_G1_PREPARE msm_g1_bn254_fixed_prepare_dev
#define G16_MSM_G2_PREPARE msm_g2_bn254_fixed_prepare_dev
#define G16_MSM_G1_FIXED msm_g1_bn254_fixed_dev
#define G16_MSM_G2_FIXED msm_g2_bn254_fixed_dev
#define G16_FQ_TO_MONT fq_bn254_canon_to_mont_dev
#define G16_FQ_TO_CANON fq_bn254_mont_to_canon_dev
#define G16_JSON_CURVE "BN128"
#define G16_FN(name) name
#include "groth16_impl.hip.h"
#undef FRN_S
#undef FRN_ROOT
#undef G16_CW
#undef G16_MSM_G1
#undef G16_MSM_G2
#undef G16_MSM_G1_TABLE_BYTES
#undef G16_MSM_G2_TABLE_BYTES
#undef G16_MSM_G1_PREPARE
#undef G16_MSM_G2_PREPARE
#undef G16_MSM_G1_FIXED
#undef G16_MSM_G2_FIXED
#undef G16_FQ_TO_MONT
#undef G16_FQ_TO_CANON
#undef G16_JSON_CURVE
}  // namespace bn254fr

namespace bls12381fr {
#define ZK_FR29_FIELD 381
#include "fr29_consts.hip.h"
#include "fe29_impl.hip.h"
#define FRN_S 32
#define FRN_ROOT 0x5f0e466au, 0xb9b58d8cu, 0x1819d7ecu, 0x5b1b4c80u, 0x52a31e64u, 0x0af53ae3u, 0x19e9b27bu, 0x5bf3addau   // 7^((r-1)/2^32) * 2^256
#include "frntt_impl.hip.h"
#define G16_CW 12
#define G16_MSM_G1 msm_g1_bls12_381_dev
#define G16_MSM_G2 msm_g2_bls12_381_dev
#define G16_MSM_G1_TABLE_BYTES msm_g1_bls12_381_fixed_table_bytes
#define G16_MSM_G2_TABLE_BYTES msm_g2_bls12_381_fixed_table_bytes
#define G16_MSM_G1_PREPARE msm_g1_bls12_381_fixed_prepare_dev
#define G16_MSM_G2_PREPARE msm_g2_bls12_381_fixed_prepare_dev
#define G16_MSM_G1_FIXED msm_g1_bls12_381_fixed_dev
#define G16_MSM_G2_FIXED msm_g2_bls12_381_fixed_dev
#define G16_FQ_TO_MONT fq_bls12_381_canon_to_mont_dev
#define G16_FQ_TO_CANON fq_bls12_381_mont_to_canon_dev
#define G16_JSON_CURVE "BLS12381"
#include "groth16_impl.hip.h"
}  // namespace bls12381fr

void fr_bn254_ntt_dev(u64* d, int logn, bool inverse, bool coset, hipStream_t st) { bn254fr::ntt_dev(d, logn, inverse, coset, st); }
void fr_bls12_381_ntt_dev(u64* d, int logn, bool inverse, bool coset, hipStream_t st) { bls12381fr::ntt_dev(d, logn, inverse, coset, st); }
void fr_bn254_quotient_dev(u64* a, const u64* b, const u64* c, int logn, hipStream_t st) { bn254fr::quotient_dev(a, b, c, logn, st); }
void fr_bls12_381_quotient_dev(u64* a, const u64* b, const u64* c, int logn, hipStream_t st) { bls12381fr::quotient_dev(a, b, c, logn, st); }

static const u32 R_BN254[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static const u32 R_BLS12_381[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};

static bool curve_is_bls(const char* curve) {
    // the reference's curve_type strings (groth16/src/api.rs:148-204)
    const std::string c = curve ? curve : "";
    if (c == "BN128") return false;
    if (c == "BLS12381") return true;
    throw std::runtime_error("groth16: unknown curve \"" + c + "\" (BN128 | BLS12381)");
}

Groth16Setup* groth16_setup_new(const char* curve, const void* r1cs, size_t r1cs_len, const void* params, size_t params_len) {
    const bool bls = curve_is_bls(curve);
    ZK_REQUIRE(r1cs && params, "groth16: null input");
    std::vector<u32> mod(bls ? R_BLS12_381 : R_BN254, (bls ? R_BLS12_381 : R_BN254) + 8);
    const g16::R1cs rc = g16::parse_r1cs((const uint8_t*)r1cs, r1cs_len, mod);
    const g16::Params pk = g16::parse_params((const uint8_t*)params, params_len, bls ? 48 : 32);
    Groth16Setup* s = bls ? bls12381fr::setup_new(rc, pk) : bn254fr::setup_new(rc, pk);
    s->curve = curve; s->modulus = mod; s->proof_words = bls ? 96 : 64;
    return s;
}

// reader.rs:86-137 load_witness_from_bin_reader: header checks, then n x 32 B little-endian canonical values
void groth16_wtns_payload(const void* wtns, size_t len, const char* curve, uint64_t* offset, uint64_t* n) {
    const bool bls = curve_is_bls(curve);
    g16::Reader rd{(const uint8_t*)wtns, len, 0, "wtns"};
    if (std::memcmp(rd.take(4), "wtns", 4) != 0) throw std::runtime_error("wtns: Invalid file header");
    if (rd.u32le() > 2) throw std::runtime_error("wtns: unsupported file version");
    if (rd.u32le() != 2) throw std::runtime_error("wtns: invalid num sections");
    if (rd.u32le() != 1) throw std::runtime_error("wtns: invalid section type");
    if (rd.u64le() != 4 + 32 + 4) throw std::runtime_error("wtns: invalid section len");
    if (rd.u32le() != 32) throw std::runtime_error("wtns: invalid field byte size");
    if (std::memcmp(rd.take(32), bls ? R_BLS12_381 : R_BN254, 32) != 0) throw std::runtime_error("wtns: invalid curve prime");
    const uint32_t cnt = rd.u32le();
    if (rd.u32le() != 2) throw std::runtime_error("wtns: invalid section type");
    if (rd.u64le() != (uint64_t)cnt * 32) throw std::runtime_error("wtns: Invalid witness section size");
    rd.need((size_t)cnt * 32);
    *offset = rd.o; *n = cnt;
}

}  // namespace zk
