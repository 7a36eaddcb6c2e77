// stark_verify (starky/src/stark_verify.rs:20-250) and FRI::verify (fri.rs:187-297) behind zk_stark_verify / zk_stark_verify_with:
// what the reference's stark_prove runs on its own proof before it writes anything (prove.rs:124-132).
//
// A verifier has no data-parallel part worth a kernel of its own -- two short straight-line programs (verifier_code at xi, verifier_query_code
// once per query), a dozen small inverse transforms of FRI groups -- except for its hashing: the Fiat-Shamir sponge, the LinearHash of every
// opened row and the walk up every Merkle path.  Those run in the library's HIP kernels (the prover's transcript, linearhash_rows_dev, a
// path kernel of 16 lanes per path, the scalar-field Poseidon for BN128 / BLS12381 trees): one upload, one launch per tree, one read-back of all
// implied roots.  The arithmetic around them is scalar work over a few hundred field elements and stays on the host, in the order of the reference.
//
// Input = the zkin JSON the prover wrote (serializer.rs:146-261) + the setup's StarkInfo / Program / StarkStruct + the root of the constants.
// Result: 1 accepted, 0 rejected (zk_last_error() says which check), -1 malformed input or a device error.
#include "zk_internal.h"
#include "../../include/zkgpu.h"
#include "json_min.h"
#include <array>
#include <atomic>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace zk;

namespace {

struct Reject { std::string why; };                     // a well-formed proof that does not verify

inline const uint64_t* C(const u64* p) { return reinterpret_cast<const uint64_t*>(p); }
inline uint64_t* M(u64* p) { return reinterpret_cast<uint64_t*>(p); }
void ck(int rc) { if (rc != 0) throw Error(zk_last_error()); }

// ---- Goldilocks and its cubic extension on the host (f3g.rs:207-235, 323-449); a base-field value is (a, 0, 0) ----
inline u64 hadd(u64 a, u64 b) { const unsigned __int128 s = (unsigned __int128)a + b; return (u64)(s >= GL_P ? s - GL_P : s); }
inline u64 hsub(u64 a, u64 b) { return a >= b ? a - b : a + (GL_P - b); }
inline u64 hneg(u64 a) { return a ? GL_P - a : 0; }
using gl::hmul; using gl::hinv; using gl::hpow; using gl::hroot;
struct F3 { u64 v[3]; };
inline F3 f3(u64 a, u64 b = 0, u64 c = 0) { return F3{{a, b, c}}; }
inline F3 operator+(const F3& a, const F3& b) { return f3(hadd(a.v[0], b.v[0]), hadd(a.v[1], b.v[1]), hadd(a.v[2], b.v[2])); }
inline F3 operator-(const F3& a, const F3& b) { return f3(hsub(a.v[0], b.v[0]), hsub(a.v[1], b.v[1]), hsub(a.v[2], b.v[2])); }
inline F3 operator*(const F3& a, const F3& b) {         // f3g.rs:420-430
    const u64 A = hmul(hadd(a.v[0], a.v[1]), hadd(b.v[0], b.v[1])), B = hmul(hadd(a.v[0], a.v[2]), hadd(b.v[0], b.v[2]));
    const u64 Cc = hmul(hadd(a.v[1], a.v[2]), hadd(b.v[1], b.v[2]));
    const u64 D = hmul(a.v[0], b.v[0]), E = hmul(a.v[1], b.v[1]), F = hmul(a.v[2], b.v[2]), G = hsub(D, E);
    return f3(hsub(hadd(Cc, G), F), hsub(hsub(hsub(hadd(A, Cc), E), E), D), hsub(B, G));
}
inline F3 operator*(const F3& a, u64 k) { return f3(hmul(a.v[0], k), hmul(a.v[1], k), hmul(a.v[2], k)); }
inline bool operator==(const F3& a, const F3& b) { return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2]; }   // _eq, f3g.rs:95-103
inline bool is_base(const F3& a) { return a.v[1] == 0 && a.v[2] == 0; }
F3 f3_inv(const F3& x) {                                 // f3g.rs:207-235; a base-field value inverts in the base field (:415-449 of field_gl.rs)
    if (is_base(x)) return f3(hinv(x.v[0]));
    const u64 a = x.v[0], b = x.v[1], c = x.v[2];
    const u64 aa = hmul(a, a), ac = hmul(a, c), ba = hmul(b, a), bb = hmul(b, b), bc = hmul(b, c), cc = hmul(c, c);
    const u64 aaa = hmul(aa, a), aac = hmul(aa, c), abc = hmul(ba, c), abb = hmul(ba, b), acc = hmul(ac, c), bbb = hmul(bb, b), bcc = hmul(bc, c), ccc = hmul(cc, c);
    u64 t = hneg(aaa);
    t = hsub(t, aac); t = hsub(t, aac); t = hadd(t, abc); t = hadd(t, abc); t = hadd(t, abc); t = hadd(t, abb);
    t = hsub(t, acc); t = hsub(t, bbb); t = hadd(t, bcc); t = hsub(t, ccc);
    const u64 ti = hinv(t);
    const u64 i1 = hmul(hsub(hadd(hadd(hsub(hsub(hneg(aa), ac), ac), bc), bb), cc), ti);
    const u64 i2 = hmul(hsub(ba, cc), ti);
    const u64 i3 = hmul(hadd(hadd(hneg(bb), ac), cc), ti);
    return f3(i1, i2, i3);
}
F3 f3_pow(F3 a, u64 e) { F3 r = f3(1); while (e) { if (e & 1) r = r * a; a = a * a; e >>= 1; } return r; }

// FFT::fft / ifft over F3G (fft.rs:39-83), the sizes of a FRI group (<= 2^11 here) and of the last polynomial
std::vector<F3> f3_fft(const std::vector<F3>& p) {
    const size_t n = p.size();
    if (n <= 1) return p;
    u32 bits = 0; while ((1ull << bits) < n) ++bits;
    if ((1ull << bits) != n) throw Reject{"FRI group whose size is not a power of two"};
    std::vector<F3> buf(n);
    for (size_t i = 0; i < n; ++i) { size_t r = 0; for (u32 k = 0; k < bits; ++k) r |= ((i >> k) & 1) << (bits - 1 - k); buf[r] = p[i]; }
    for (u32 s = 1; s <= bits; ++s) {
        const size_t m = 1ull << s, h = m >> 1;
        const u64 winc = hroot(s);
        for (size_t k = 0; k < n; k += m) {
            u64 w = 1;
            for (size_t j = 0; j < h; ++j) { const F3 t = buf[k + j + h] * w, u = buf[k + j]; buf[k + j] = u + t; buf[k + j + h] = u - t; w = hmul(w, winc); }
        }
    }
    return buf;
}
std::vector<F3> f3_ifft(const std::vector<F3>& p) {
    const std::vector<F3> q = f3_fft(p);
    const size_t n = p.size();
    if (n == 0) return q;
    const u64 ninv = hinv((u64)n % GL_P);
    std::vector<F3> r(n);
    r[0] = q[0] * ninv;
    for (size_t i = 1; i < n; ++i) r[i] = q[n - i] * ninv;
    return r;
}
F3 eval_pol(const std::vector<F3>& p, const F3& x) {       // polutils.rs:13-23
    if (p.empty()) return f3(0);
    F3 r = p.back();
    for (size_t i = p.size() - 1; i-- > 0;) r = r * x + p[i];
    return r;
}

u64 parse_word(const JVal& v) {                           // a Goldilocks word as the serializer prints it: a decimal string
    const std::string& s = v.kind == JVal::Str ? v.s : v.kind == JVal::Num ? v.s : throw Error("zkin: number expected");
    if (s.empty() || s.size() > 20) throw Error("zkin: bad number");
    unsigned __int128 x = 0;
    for (char c : s) { if (c < '0' || c > '9') throw Error("zkin: bad number"); x = x * 10 + (unsigned)(c - '0'); }
    if (x >> 64) throw Error("zkin: number out of range");
    return (u64)(x % GL_P);                               // FGL::from(u64) reduces
}
u64 parse_pil_number(const std::string& s) {              // types.rs:221-233
    bool neg = !s.empty() && s[0] == '-';
    size_t i = neg ? 1 : 0;
    unsigned __int128 v = 0;
    if (s.size() > i + 1 && s[i] == '0' && (s[i + 1] == 'x' || s[i + 1] == 'X')) {
        for (i += 2; i < s.size(); ++i) {
            const char c = s[i];
            const int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
            if (d < 0) throw Error("bad PIL number " + s);
            v = (v * 16 + d) % GL_P;
        }
    } else for (; i < s.size(); ++i) { if (s[i] < '0' || s[i] > '9') throw Error("bad PIL number " + s); v = (v * 10 + (s[i] - '0')) % GL_P; }
    const u64 r = (u64)v;
    return neg && r ? GL_P - r : r;
}

enum Hash { H_GL, H_BN128, H_BLS12381 };

// the sponge of the proof's hash type through the library's own transcripts (device permutations)
struct Sponge {
    Hash h; zk_transcript_t* gl = nullptr; zk_bn128_transcript_t* bn = nullptr; zk_bls12381_transcript_t* bls = nullptr;
    explicit Sponge(Hash hh) : h(hh) {
        if (h == H_GL) gl = zk_transcript_new(); else if (h == H_BN128) bn = zk_bn128_transcript_new(); else bls = zk_bls12381_transcript_new();
        if (!gl && !bn && !bls) throw Error(zk_last_error());
    }
    Sponge(const Sponge&) = delete; Sponge& operator=(const Sponge&) = delete;
    ~Sponge() { if (gl) zk_transcript_free(gl); if (bn) zk_bn128_transcript_free(bn); if (bls) zk_bls12381_transcript_free(bls); }
    void put_words(const u64* w, size_t n) {              // n one-word elements (publics, evaluations, the last polynomial)
        if (!n) return;
        if (gl) { ck(zk_transcript_put(gl, C(w), n)); return; }
        for (size_t i = 0; i < n; ++i) ck(bn ? zk_bn128_transcript_put(bn, C(w + i), 1) : zk_bls12381_transcript_put(bls, C(w + i), 1));
    }
    void put_root(const u64 r[4]) { ck(gl ? zk_transcript_put(gl, C(r), 4) : bn ? zk_bn128_transcript_put(bn, C(r), 4) : zk_bls12381_transcript_put(bls, C(r), 4)); }
    F3 get_field() {
        u64 o[3];
        ck(gl ? zk_transcript_get_field(gl, M(o)) : bn ? zk_bn128_transcript_get_field(bn, M(o)) : zk_bls12381_transcript_get_field(bls, M(o)));
        return f3(o[0], o[1], o[2]);
    }
    std::vector<u64> get_permutations(u32 n, u32 nbits) {
        std::vector<u64> o(std::max<u32>(1, n));
        ck(gl ? zk_transcript_get_permutations(gl, n, nbits, M(o.data())) : bn ? zk_bn128_transcript_get_permutations(bn, n, nbits, M(o.data()))
                                                                                : zk_bls12381_transcript_get_permutations(bls, n, nbits, M(o.data())));
        o.resize(n);
        return o;
    }
};

struct Opening { std::vector<u64> row; std::vector<u64> path; u32 depth = 0; };   // path: depth x 4 words (GL) or depth x 16 x 4 (scalar field)
struct TreeOpenings { u64 root[4]; std::vector<Opening> q; };

struct Proof {
    Hash h = H_GL;
    u64 root[4][4];
    std::vector<F3> evals; std::vector<u64> publics;
    TreeOpenings s0[5];                                    // tree1..4, constants (root filled by the caller)
    std::vector<TreeOpenings> steps;                       // FRI steps 1..
    std::vector<F3> last;
};

void parse_digest(const JVal& v, Hash h, u64 out[4]) {     // digest.rs:84-112 read backwards
    if (h == H_GL) {
        if (v.kind == JVal::Arr) { if (v.size() != 4) throw Error("zkin: a digest has 4 words"); for (int i = 0; i < 4; ++i) out[i] = parse_word(v.at(i)); }
        else { out[0] = parse_word(v); out[1] = out[2] = out[3] = 0; }
        return;
    }
    if (v.kind != JVal::Str || !fr_digest_from_dec(h == H_BLS12381, v.s, out)) throw Error("zkin: a digest is a canonical scalar-field element in decimal");
}
Opening parse_opening(const JVal& row, const JVal& sib, Hash h) {
    Opening o;
    if (row.kind != JVal::Arr || sib.kind != JVal::Arr) throw Error("zkin: opening: arrays expected");
    for (const JVal& w : row.arr) o.row.push_back(parse_word(w));
    o.depth = (u32)sib.size();
    if (o.depth > 64) throw Error("zkin: path too long");
    for (const JVal& lvl : sib.arr) {
        if (h == H_GL) {
            if (lvl.kind != JVal::Arr || lvl.size() != 4) throw Error("zkin: a sibling is a digest of 4 words");
            for (const JVal& w : lvl.arr) o.path.push_back(parse_word(w));
        } else {
            if (lvl.kind != JVal::Arr || lvl.size() != 16) throw Error("zkin: a level of a 16-ary path holds 16 nodes");
            for (const JVal& w : lvl.arr) { u64 d[4]; parse_digest(w, h, d); o.path.insert(o.path.end(), d, d + 4); }
        }
    }
    return o;
}
F3 parse_f3(const JVal& v) {                               // serializer.rs:21-39: one string (dim 1) or three
    if (v.kind == JVal::Arr) { if (v.size() != 3) throw Error("zkin: an extension value has 3 words"); return f3(parse_word(v.at(0)), parse_word(v.at(1)), parse_word(v.at(2))); }
    return f3(parse_word(v));
}
void parse_tree(const JVal& Z, const std::string& vals, const std::string& sibs, Hash h, u32 nq, TreeOpenings& t) {
    const JVal &V = Z.at(vals), &S = Z.at(sibs);
    if (V.size() != nq || S.size() != nq) throw Reject{vals + ": one opening per query expected"};
    for (u32 i = 0; i < nq; ++i) t.q.push_back(parse_opening(V.at(i), S.at(i), h));
}
Proof parse_proof(const JVal& Z, Hash h, u32 nq, size_t n_steps) {
    Proof P; P.h = h;
    const char* rn[4] = {"root1", "root2", "root3", "root4"};
    for (int j = 0; j < 4; ++j) parse_digest(Z.at(rn[j]), h, P.root[j]);
    for (const JVal& e : Z.at("evals").arr) P.evals.push_back(parse_f3(e));
    for (const JVal& p : Z.at("publics").arr) P.publics.push_back(parse_word(p));
    const char* nm[5] = {"1", "2", "3", "4", "C"};
    for (int j = 0; j < 5; ++j) parse_tree(Z, std::string("s0_vals") + nm[j], std::string("s0_siblings") + nm[j], h, nq, P.s0[j]);
    P.steps.resize(n_steps > 0 ? n_steps - 1 : 0);
    for (size_t si = 1; si < n_steps; ++si) {
        const std::string s = "s" + std::to_string(si);
        if (!Z.find(s + "_root")) throw Reject{"the proof has fewer FRI steps than the starkStruct"};   // fri.rs:196 assert_eq
        parse_digest(Z.at(s + "_root"), h, P.steps[si - 1].root);
        parse_tree(Z, s + "_vals", s + "_siblings", h, nq, P.steps[si - 1]);
    }
    if (Z.find("s" + std::to_string(n_steps) + "_root")) throw Reject{"the proof has more FRI steps than the starkStruct"};
    for (const JVal& e : Z.at("finalPol").arr) P.last.push_back(parse_f3(e));
    return P;
}

// execute_code (stark_verify.rs:138-250)
struct ExecCtx {
    const std::vector<F3>* evals; const std::vector<u64>* publics; const F3* challenge;
    const std::vector<u64>* tree[4] = {nullptr, nullptr, nullptr, nullptr}; const std::vector<u64>* consts = nullptr;
    F3 Z, Zp, xdiv, xdivw;
};
F3 execute_code(const JVal& code, const ExecCtx& c) {
    if (code.size() == 0) throw Error("verifier program is empty");
    std::vector<F3> tmp; std::vector<char> set;
    auto get = [&](const JVal& r) -> F3 {
        const std::string& t = r.at("type_").str();
        auto id = [&] { return (size_t)r.at("id").u64(); };
        if (t == "tmp") { const size_t i = id(); if (i >= tmp.size() || !set[i]) throw Error("verifier program reads a temporary before it is written"); return tmp[i]; }
        if (t.size() == 5 && t.compare(0, 4, "tree") == 0 && t[4] >= '1' && t[4] <= '4') {
            const std::vector<u64>* a = c.tree[t[4] - '1'];
            if (!a) throw Error("verifier program reads a tree outside a query");
            const size_t p = (size_t)r.at("tree_pos").u64(), dim = (size_t)r.at("dim").u64();
            if (dim != 1 && dim != 3) throw Error("Invalid dimension");
            if (p + dim > a->size()) throw Reject{"an opened row is shorter than the verifier program expects"};
            return dim == 1 ? f3((*a)[p]) : f3((*a)[p], (*a)[p + 1], (*a)[p + 2]);
        }
        if (t == "const") { if (!c.consts) throw Error("verifier program reads the constants outside a query"); if (id() >= c.consts->size()) throw Reject{"the opened constants row is too short"}; return f3((*c.consts)[id()]); }
        if (t == "eval") { if (id() >= c.evals->size()) throw Reject{"the proof holds fewer evaluations than the verifier program reads"}; return (*c.evals)[id()]; }
        if (t == "number") return f3(parse_pil_number(r.at("value").str()));
        if (t == "public") { if (id() >= c.publics->size()) throw Reject{"the proof holds fewer publics than the verifier program reads"}; return f3((*c.publics)[id()]); }
        if (t == "challenge") { if (id() >= 8) throw Error("challenge id out of range"); return c.challenge[id()]; }
        if (t == "xDivXSubXi") return c.xdiv;
        if (t == "xDivXSubWXi") return c.xdivw;
        if (t == "x") return c.challenge[7];
        if (t == "Z") return r.at("prime").boolean() ? c.Zp : c.Z;
        throw Error("Invalid reference type, get: " + t);
    };
    for (const JVal& ci : code.arr) {
        const std::string& op = ci.at("op").str();
        const JVal& src = ci.at("src");
        auto need = [&](size_t n) { if (src.size() < n) throw Error("verifier instruction with too few sources"); };
        F3 r;
        if (op == "add") { need(2); r = get(src.at(0)) + get(src.at(1)); }
        else if (op == "sub") { need(2); r = get(src.at(0)) - get(src.at(1)); }
        else if (op == "mul") { need(2); r = get(src.at(0)) * get(src.at(1)); }
        else if (op == "muladd") { need(3); r = get(src.at(0)) * get(src.at(1)) + get(src.at(2)); }
        else if (op == "copy") { need(1); r = get(src.at(0)); }
        else throw Error("Invalid op: " + op);
        const JVal& d = ci.at("dest");
        if (d.at("type_").str() != "tmp") throw Error("Invalid reference type set: " + d.at("type_").str());
        const size_t i = (size_t)d.at("id").u64();
        if (i >= (1u << 24)) throw Error("temporary id out of range");
        if (i >= tmp.size()) { tmp.resize(i + 1, f3(0)); set.resize(i + 1, 0); }
        tmp[i] = r; set[i] = 1;
    }
    return get(code.arr.back().at("dest"));
}

// The root every opening implies (verify_group_proof: merklehash.rs:440-453, merklehash_bn128.rs:260-269), all trees at once, on the
// device.  Scalar-field trees: the reference's merkle_calculate_root_from_proof (merklehash_bn128.rs:108-128) hashes the 16 nodes of each
// level of the path but never looks for the value carried up among them, so the root it returns is the hash of the LAST level's 16 nodes
// (or the leaf digest for an empty path): rows and lower levels are not bound to the root.  The reference only runs that check on its
// own proof; this verifier takes untrusted zkin, so by default it is STRICT: at every level the node at position idx & 15 must equal the
// value carried up (the leaf digest first), then idx >>= 4 -- every honest proof passes, and proofs the reference would wave through
// with forged rows are rejected.  g_reference_compat (zk_stark_verify_set_reference_compat) restores the reference's behaviour for the
// parity tests against the restated verifier.
thread_local int g_reference_compat = 0;   // per calling thread: a parity test on one thread must not weaken verifications running on others (advisor finding, round 5)
struct PathRef { const Opening* o; u64 idx; const u64* want; const char* what; };
void check_paths(Hash h, const std::vector<std::vector<PathRef>>& trees /* paths of one tree share the row width */) {
    size_t n = 0; u32 max_depth = 1;
    for (auto& t : trees) { n += t.size(); for (auto& p : t) max_depth = std::max(max_depth, p.o->depth); }
    if (n == 0) return;
    hipStream_t st = cur_stream();                       // (the scalar-field sponges move the thread to the null stream: come back)
    on_stream(st);
    size_t levels = 0;                                  // scalar fields: every level of every path is hashed (16 nodes of 4 words each)
    for (auto& t : trees) for (auto& p : t) levels += p.o->depth;
    std::vector<u64> h_paths(h == H_GL ? n * max_depth * 4 : std::max<size_t>(1, levels) * 64, 0), h_idx(n);
    std::vector<u32> h_depth(n);
    DevBuf d_leaves, d_paths, d_idx, d_depth, d_roots, d_zero;
    d_leaves.reserve(n * 32); d_roots.reserve(std::max(n, levels) * 64); d_paths.reserve(h_paths.size() * 8); d_idx.reserve(n * 8); d_depth.reserve(n * 4);
    std::vector<std::unique_ptr<DevBuf>> rows_keep;
    size_t k = 0, lv = 0;
    for (auto& t : trees) {
        if (t.empty()) continue;
        const size_t w = t[0].o->row.size();
        std::vector<u64> rows(std::max<size_t>(1, w * t.size()));
        for (size_t i = 0; i < t.size(); ++i) {
            if (t[i].o->row.size() != w) throw Reject{std::string(t[i].what) + ": rows of one tree differ in width"};
            memcpy(rows.data() + i * w, t[i].o->row.data(), w * 8);
            const Opening& o = *t[i].o;
            h_idx[k + i] = t[i].idx; h_depth[k + i] = o.depth;
            if (h == H_GL) memcpy(h_paths.data() + (k + i) * max_depth * 4, o.path.data(), (size_t)o.depth * 32);
            else if (o.depth) { memcpy(h_paths.data() + lv * 64, o.path.data(), (size_t)o.depth * 512); lv += o.depth; }
        }
        rows_keep.emplace_back(new DevBuf); DevBuf& d_rows = *rows_keep.back(); d_rows.reserve(rows.size() * 8);
        h2d_sync(d_rows.p, rows.data(), rows.size() * 8);
        if (h == H_GL) linearhash_rows_dev(d_rows.u(), (u32)w, t.size(), d_leaves.u() + 4 * k, st);
        else fr_linearhash_rows_dev(h == H_BLS12381, d_rows.u(), (u32)w, t.size(), d_leaves.u() + 4 * k, st);
        k += t.size();
    }
    h2d_sync(d_paths.p, h_paths.data(), h_paths.size() * 8);
    std::vector<u64> got(8 * std::max(n, levels)), leaves;   // scalar fields: two words of the permutation per level, the hash is word 0 (BN128) or 1 (BLS12-381)
    const size_t stride = h == H_GL ? 4 : 8, pick = h == H_BLS12381 ? 4 : 0;
    if (h == H_GL) {
        h2d_sync(d_idx.p, h_idx.data(), n * 8); h2d_sync(d_depth.p, h_depth.data(), n * 4);
        merkle_roots_from_paths_dev(d_leaves.u(), d_paths.u(), (const u32*)d_depth.p, d_idx.u(), (u32)n, max_depth, d_roots.u(), st);
    } else {
        d_zero.reserve(32); ZK_HIP(hipMemsetAsync(d_zero.p, 0, 32, st));
        if (levels) fr_hash16_dev(h == H_BLS12381, d_paths.u(), levels, d_zero.u(), d_roots.u(), st);
        leaves.resize(4 * n); d2h_sync(leaves.data(), d_leaves.p, n * 32);
    }
    d2h_sync(got.data(), d_roots.p, (h == H_GL ? n : levels) * stride * 8);
    const bool strict = g_reference_compat == 0;
    k = 0; lv = 0;
    for (auto& t : trees)
        for (auto& p : t) {
            const u64* r;
            if (h == H_GL) r = got.data() + stride * k;
            else {                                       // walk the 16-ary path: leaf digest -> level 0 -> ... -> root
                r = leaves.data() + 4 * k;
                u64 idx = p.idx;
                for (u32 L = 0; L < p.o->depth; ++L, ++lv) {
                    if (strict && memcmp(h_paths.data() + lv * 64 + 4 * (idx & 15), r, 32) != 0)
                        throw Reject{std::string("FRIVerifierFailed: ") + p.what + ": level " + std::to_string(L) + " of the path does not hold the value carried up (strict check; "
                                     "merklehash_bn128.rs:108-128 does not make it)"};
                    r = got.data() + stride * lv + pick;
                    idx >>= 4;
                }
            }
            if (memcmp(r, p.want, 32) != 0) throw Reject{std::string("FRIVerifierFailed: ") + p.what + " does not open to its root"};
            ++k;
        }
}

bool verify(const JVal& info, const JVal& prog, const JVal& ss, const u64 const_root[4], const char* zkin_json) {
    const u32 nbits = (u32)ss.at("nBits").u64(), nbits_ext = (u32)ss.at("nBitsExt").u64(), nq = (u32)ss.at("nQueries").u64();
    ZK_REQUIRE(nbits >= 1 && nbits <= nbits_ext && nbits_ext <= 32, "bad nBits / nBitsExt");
    const std::string& ht = ss.at("verificationHashType").str();
    ZK_REQUIRE(ht == "GL" || ht == "BN128" || ht == "BLS12381", "verificationHashType must be GL, BN128 or BLS12381");
    const Hash h = ht == "GL" ? H_GL : ht == "BN128" ? H_BN128 : H_BLS12381;
    std::vector<u32> steps;
    for (const JVal& s : ss.at("steps").arr) steps.push_back((u32)s.at("nBits").u64());
    ZK_REQUIRE(!steps.empty() && steps[0] <= 32, "starkStruct without FRI steps");
    for (size_t i = 1; i < steps.size(); ++i) ZK_REQUIRE(steps[i] <= steps[i - 1], "FRI steps must not grow");
    const JVal Z = JParser::parse(zkin_json);
    Proof P = parse_proof(Z, h, nq, steps.size());
    memcpy(P.s0[4].root, const_root, 32);
    for (int j = 0; j < 4; ++j) memcpy(P.s0[j].root, P.root[j], 32);

    // ---- the challenges (stark_verify.rs:28-61)
    Sponge tr(h);
    F3 ch[8]; for (F3& c : ch) c = f3(0);
    tr.put_words(P.publics.data(), P.publics.size());
    tr.put_root(P.root[0]); ch[0] = tr.get_field(); ch[1] = tr.get_field();
    tr.put_root(P.root[1]); ch[2] = tr.get_field(); ch[3] = tr.get_field();
    tr.put_root(P.root[2]); ch[4] = tr.get_field();
    tr.put_root(P.root[3]); ch[7] = tr.get_field();
    { std::vector<u64> w; for (const F3& e : P.evals) w.insert(w.end(), e.v, e.v + 3); tr.put_words(w.data(), w.size()); }
    ch[5] = tr.get_field(); ch[6] = tr.get_field();

    // ---- Q(xi) Z(xi) == C(xi) (:63-84)
    const u64 N = 1ull << nbits, w_n = hroot(nbits);
    const F3 x_n = f3_pow(ch[7], N);
    ExecCtx c0; c0.evals = &P.evals; c0.publics = &P.publics; c0.challenge = ch;
    c0.Z = x_n - f3(1); c0.Zp = f3_pow(ch[7] * w_n, N) - f3(1); c0.xdiv = c0.xdivw = f3(0);
    const F3 res = execute_code(prog.at("verifier_code").at("first"), c0);
    F3 x_acc = f3(1), q = f3(0);
    const u64 q_deg = info.at("q_deg").u64();
    const JVal& qs = info.at("qs");
    for (u64 i = 0; i < q_deg; ++i) {
        const u64 want = qs.at(i).u64();
        const JVal* hit = nullptr;
        for (const JVal& e : info.at("ev_idx").at("cm").arr) if (e.at(0).at(0).u64() == 0 && e.at(0).at(1).u64() == want) { hit = &e; break; }
        if (!hit) throw Error("ev_idx: no evaluation of a quotient piece");
        const size_t k = (size_t)hit->at(1).u64();
        if (k >= P.evals.size()) throw Reject{"the proof holds fewer evaluations than the starkinfo names"};
        q = q + x_acc * P.evals[k];
        x_acc = x_acc * x_n;
    }
    if (!(res == q * c0.Z)) throw Reject{"Q != C * P at xi (Eq. 30 of the eSTARK paper)"};

    // ---- FRI::verify (fri.rs:187-297): the folding challenges, then the query indices
    const size_t n_steps = steps.size();
    std::vector<F3> special_x;
    for (size_t si = 0; si < n_steps; ++si) {
        special_x.push_back(tr.get_field());
        if (si + 1 < n_steps) tr.put_root(P.steps[si].root);
        else { std::vector<u64> w; for (const F3& e : P.last) w.insert(w.end(), e.v, e.v + 3); tr.put_words(w.data(), w.size()); }
    }
    std::vector<u64> ys = tr.get_permutations(nq, steps[0]);
    // untrusted input: the last polynomial has exactly 2^steps.last values (the reference indexes it blindly, fri.rs:262-276) ...
    if (P.last.size() != (1ull << steps.back())) throw Reject{"the last polynomial does not hold 2^steps.last values"};

    // every opening against its root: the five trees at ys, the folded polynomials' trees at the reduced indices
    {
        std::vector<std::vector<PathRef>> trees(5 + (n_steps - 1));
        static const char* const what[5] = {"tree1", "tree2", "tree3", "tree4", "the constants' tree"};
        for (int j = 0; j < 5; ++j) for (u32 i = 0; i < nq; ++i) trees[j].push_back(PathRef{&P.s0[j].q[i], ys[i], P.s0[j].root, what[j]});
        std::vector<u64> yr = ys;
        for (size_t si = 1; si < n_steps; ++si) {
            for (u64& y : yr) y &= (1ull << steps[si]) - 1;
            for (u32 i = 0; i < nq; ++i) trees[4 + si].push_back(PathRef{&P.steps[si - 1].q[i], yr[i], P.steps[si - 1].root, "a FRI step"});
        }
        // ... and every path is as deep as the tree its StarkStruct implies (binary GL trees: log2 height; 16-ary scalar-field trees: ceil(log16))
        auto want_depth = [&](u32 height_bits) -> u32 { return h == H_GL ? height_bits : (height_bits + 3) / 4; };
        for (size_t j = 0; j < trees.size(); ++j) {
            const u32 hb = j < 5 ? nbits_ext : steps[j - 4];
            for (const PathRef& pr : trees[j])
                if (pr.o->depth != want_depth(hb)) throw Reject{std::string(pr.what) + ": the Merkle path has " + std::to_string(pr.o->depth) + " levels, the tree has " + std::to_string(want_depth(hb))};
        }
        check_paths(h, trees);
    }

    u32 pol_bits = nbits_ext;
    u64 shift = 49;
    const u64 w_ext = hroot(nbits_ext);
    const JVal& qcode = prog.at("verifier_query_code").at("first");
    for (size_t si = 0; si < n_steps; ++si) {
        if (steps[si] > pol_bits) throw Error("FRI steps must not grow");
        const u32 reduction_bits = pol_bits - steps[si];
        for (u32 i = 0; i < nq; ++i) {
            std::vector<F3> group;
            if (si == 0) {                                 // check_query (stark_verify.rs:86-133)
                ExecCtx cq = c0;
                for (int j = 0; j < 4; ++j) cq.tree[j] = &P.s0[j].q[i].row;
                cq.consts = &P.s0[4].q[i].row;
                const F3 x = f3(hmul(49, hpow(w_ext, ys[i])));
                cq.xdiv = x * f3_inv(x - ch[7]);
                cq.xdivw = x * f3_inv(x - ch[7] * w_n);
                group.push_back(execute_code(qcode, cq));
            } else {                                       // split3 (fri.rs:228-238)
                const std::vector<u64>& row = P.steps[si - 1].q[i].row;
                if (row.size() % 3) throw Reject{"a FRI opening is not a list of extension values"};
                for (size_t k = 0; k + 3 <= row.size(); k += 3) group.push_back(f3(row[k], row[k + 1], row[k + 2]));
                if (group.size() != (1ull << reduction_bits)) throw Reject{"a FRI opening has the wrong group size"};
            }
            const std::vector<F3> coef = f3_ifft(group);
            const u64 sinv = hinv(hmul(shift, hpow(hroot(pol_bits), ys[i])));
            const F3 ev = eval_pol(coef, special_x[si] * sinv);
            if (si + 1 < n_steps) {
                const u64 next_groups = 1ull << steps[si + 1], gi = ys[i] / next_groups;
                const std::vector<u64>& nrow = P.steps[si].q[i].row;
                if (3 * gi + 3 > nrow.size()) throw Reject{"a FRI opening is shorter than the fold it has to contain"};
                if (!(ev == f3(nrow[3 * gi], nrow[3 * gi + 1], nrow[3 * gi + 2]))) throw Reject{"FRI: a fold does not match the next step's opening (step " + std::to_string(si + 1) + ")"};
            } else {
                if (ys[i] >= P.last.size()) throw Reject{"the last polynomial is shorter than 2^steps.last"};
                if (!(ev == P.last[ys[i]])) throw Reject{"FRI: the last fold does not match the last polynomial"};
            }
        }
        pol_bits = steps[si];
        for (u32 k = 0; k < reduction_bits; ++k) shift = hmul(shift, shift);
        if (si + 1 < n_steps) for (u64& y : ys) y %= 1ull << steps[si + 1];
    }
    // the last polynomial's degree (fri.rs:278-295)
    const u32 in_bits = nbits_ext, max_deg_bits = nbits;
    const u64 max_deg = pol_bits < in_bits - max_deg_bits ? 0 : 1ull << (pol_bits - (in_bits - max_deg_bits));
    const std::vector<F3> last_c = f3_ifft(P.last);
    for (u64 i = max_deg + 1; i < last_c.size(); ++i) if (!(last_c[i] == f3(0))) throw Reject{"FRI: the last polynomial's degree is too high"};
    return true;
}

}  // namespace

namespace zk {
// 1 accepted / 0 rejected (why says which check); throws Error on malformed input
int stark_verify_impl(const JVal& info, const JVal& prog, const JVal& ss, const u64 const_root[4], const char* zkin_json, std::string& why) {
    try { verify(info, prog, ss, const_root, zkin_json); return 1; }
    catch (const Reject& r) { why = r.why; return 0; }
    catch (const std::runtime_error& e) {
        if (dynamic_cast<const Error*>(&e)) throw;
        throw Error(std::string("zkin: ") + e.what());   // json_min's own exceptions
    }
}
}  // namespace zk

extern "C" int zk_stark_verify_set_reference_compat(int on) {
    const int old = g_reference_compat;
    g_reference_compat = on ? 1 : 0;
    return old;
}

extern "C" int zk_stark_verify_with(const char* starkinfo_program_json, const char* stark_struct_json, const uint64_t const_root[4], const char* zkin_json) {
    CallScope scope;
    try {
        if (!starkinfo_program_json || !stark_struct_json || !const_root || !zkin_json) throw Error("zk_stark_verify_with: null argument");
        const JVal root = JParser::parse(starkinfo_program_json), ss = JParser::parse(stark_struct_json);
        std::string why;
        const int ok = stark_verify_impl(root.at("starkinfo"), root.at("program"), ss, reinterpret_cast<const u64*>(const_root), zkin_json, why);
        if (!ok) set_error("stark_verify: " + why);
        return ok;
    } catch (const std::exception& e) { set_error(e.what()); return -1; }
    catch (...) { set_error("unknown error"); return -1; }
}
