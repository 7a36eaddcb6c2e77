// PIL -> prover program: the code generator of the STARK prover, host side (pure CPU, runs once per circuit).
//
// Stands behind StarkInfo::new (starky/src/starkinfo.rs:160-272) as called from StarkSetup::new
// (starky/src/stark_setup.rs:27-66): takes the compiled PIL (types.rs:134-155) and the StarkStruct and produces
// `{"starkinfo": StarkInfo, "program": Program}` in the reference's serde field names (starkinfo.rs:27-95,
// starkinfo_codegen.rs:50-89) -- exactly what zk_stark_setup_new consumes, so a caller without the Rust front end
// (tools/zkgpu_prove.py, bench.py) goes from a .pil.json to a proof inside the product.
//
//   expressionops.rs            expression constructors
//   starkinfo_codegen.rs        pil_code_gen / eval_exp / build_code / find_muladd (:296-669)
//   starkinfo.rs:274-408        public calculators, plookup step 2
//   starkinfo_Z.rs              permutation / plookup / connection grand products (step 3)
//   starkinfo_cp_prover.rs      constraint polynomial, intermediate-polynomial search (:12-291)
//   starkinfo_cp_ver.rs         verifier's constraint code, ev_map (:8-119)
//   starkinfo_fri_prover.rs / starkinfo_fri_ver.rs    FRI polynomial
//   starkinfo_map.rs            sections, offsets, operand fix-ups, dimensions (:10-640)
//
// Everything here must be deterministic and order-exact: temporaries, ev_map and section positions end up in the
// transcript order and in the circom verifier (SURVEY hard part 3).  tests/test_starkinfo_native.py compares the output
// with the test suite's independent restatement on every fixture.
#include "zk_internal.h"
#include "json_min.h"
#include "../../include/zkgpu.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <sstream>

namespace zk {
namespace sgen {

typedef long long i64;
static const char* const SECTIONS[11] = {"cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "q_2ns", "f_2ns", "tmpexp_n"};
static const char* const GLOBAL_L1 = "Global.L1";                                        // constant.rs:118

// ---- types.rs:36-60 Expression -----------------------------------------------------------------------------------
struct Expr {
    std::string op;
    i64 deg = 0;
    bool has_id = false; i64 id = 0;
    bool next = false;
    bool has_value = false; std::string value;
    bool has_values = false; std::vector<Expr> values;
    bool keep = false, keep2ns = false;          // Option<bool>: only ever None or Some(true)
    bool has_idq = false; i64 idq = 0;
    bool has_const = false; i64 const_ = 0;
};
static Expr E(const std::string& op) { Expr e; e.op = op; return e; }
static Expr E_id(const std::string& op, i64 id, bool next = false) { Expr e; e.op = op; e.has_id = true; e.id = id; e.next = next; return e; }
static Expr E2(const std::string& op, const Expr& a, const Expr& b) { Expr e; e.op = op; e.has_values = true; e.values = {a, b}; return e; }
static Expr e_add(const Expr& a, const Expr& b) { return E2("add", a, b); }
static Expr e_sub(const Expr& a, const Expr& b) { return E2("sub", a, b); }
static Expr e_mul(const Expr& a, const Expr& b) { return E2("mul", a, b); }
static Expr e_exp(i64 i, bool next = false) { return E_id("exp", i, next); }
static Expr e_cm(i64 i, bool next = false) { return E_id("cm", i, next); }
static Expr e_const(i64 i) { return E_id("const", i); }
static Expr e_q(i64 i) { return E_id("q", i); }
static Expr e_eval(i64 i) { return E_id("eval", i); }
static Expr e_challenge(int id) { return E_id("challenge", id); }                          // constant.rs:39-50
static Expr e_number(const std::string& v) { Expr e; e.op = "number"; e.has_value = true; e.value = v; return e; }
static bool is_nop(const Expr& e) { return e.op == "nop"; }
enum { CH_U = 0, CH_DEFVAL = 1, CH_GAMMA = 2, CH_BETA = 3, CH_VC = 4, CH_VF1 = 5, CH_VF2 = 6, CH_XI = 7 };

static Expr load_expr(const JVal& d) {
    Expr e;
    e.op = d.at("op").str();
    if (const JVal* v = d.find("deg")) if (!v->is_null()) e.deg = v->i64();
    if (const JVal* v = d.find("id")) if (!v->is_null()) { e.has_id = true; e.id = v->i64(); }
    if (const JVal* v = d.find("next")) if (!v->is_null()) e.next = v->boolean();
    if (const JVal* v = d.find("value")) if (!v->is_null()) { e.has_value = true; e.value = v->kind == JVal::Str ? v->s : v->s; }
    for (const char* k : {"const", "const_"})
        if (const JVal* v = d.find(k)) if (!v->is_null()) { e.has_const = true; e.const_ = v->i64(); }
    if (const JVal* v = d.find("keep")) if (!v->is_null()) e.keep = true;
    if (const JVal* v = d.find("keep2ns")) if (!v->is_null()) e.keep2ns = true;
    if (const JVal* v = d.find("idQ")) if (!v->is_null()) { e.has_idq = true; e.idq = v->i64(); }
    if (const JVal* v = d.find("values")) if (!v->is_null()) { e.has_values = true; for (auto& x : v->arr) e.values.push_back(load_expr(x)); }
    return e;
}

// ---- starkinfo_codegen.rs:50-89 ------------------------------------------------------------------------------------
struct Node {
    std::string type_;
    i64 id = 0;
    bool has_value = false; std::string value;
    i64 dim = 0; bool prime = false; i64 tree_pos = 0, p = 0, exp_id = 0;
};
static Node N(const std::string& t, i64 id = 0, bool prime = false) { Node n; n.type_ = t; n.id = id; n.prime = prime; return n; }
static Node N_num(const std::string& v) { Node n; n.type_ = "number"; n.has_value = true; n.value = v; return n; }
struct Section { std::string op; Node dest; std::vector<Node> src; };
struct Code { i64 exp_id; bool prime; std::vector<Section> code; };
struct Segment { std::vector<Section> first, i, last; i64 tmp_used = 0; };
struct Ctx {
    i64 tmp_used = 0;
    std::vector<Code> code;
    std::map<std::pair<int, i64>, bool> calculated;        // (prime?, exp id) -> flag; presence is what pil_code_gen tests
};
struct PCCtx { i64 f_exp_id = 0, t_exp_id = 0, h1_id = 0, h2_id = 0, z_id = 0, c1_id = 0, c2_id = 0, num_id = 0, den_id = 0; };
struct PolInfo { std::string section; i64 section_pos = 0, dim = 0, exp_id = 0; };
struct Identity { std::vector<i64> f, t; bool has_self = false, has_selt = false; i64 self_ = 0, selt = 0; };
struct Connection { std::vector<i64> pols, connections; };
struct Public { std::string polType, name; i64 polId = 0, idx = 0, id = 0; };

struct Pil {
    i64 nCommitments = 0, nQ = 0, nConstants = 0;
    std::vector<Public> publics;
    std::map<std::string, std::pair<i64, i64>> references;  // name -> (id, polDeg)
    i64 first_pol_deg = -1;
    std::vector<Expr> expressions;
    std::vector<i64> pol_identities;
    std::vector<Identity> plookups, permutations;
    std::vector<Connection> connections;
    std::vector<i64> cm_dims, q2exp;
};

struct Info {
    std::vector<PolInfo> var_pol_map;
    i64 n_cm1 = 0, n_cm2 = 0, n_cm3 = 0, n_cm4 = 0, n_q = 0, n_constants = 0, n_publics = 0, c_exp = 0, q_deg = 0, q_dim = 0;
    i64 fri_exp_id = 0, n_exps = 0, map_total_n = 0;
    std::vector<PCCtx> pu_ctx, pe_ctx, ci_ctx;
    std::map<i64, bool> im_exps;
    std::vector<i64> im_exps_list;
    std::map<i64, i64> im_exp2cm, exp2pol;
    std::vector<i64> qs, cm_n, cm_2ns, tmpexp_n, q_2ns, f_2ns;
    std::vector<Node> ev_map;
    std::vector<std::pair<std::pair<i64, i64>, i64>> ev_cm, ev_const;   // insertion order = serialisation order
    std::map<std::string, std::vector<i64>> map_sections;
    std::map<std::string, i64> map_sectionsN1, map_sectionsN3, map_sectionsN, map_offsets, map_deg;
};
struct Program { std::vector<Segment> publics_code; Segment step2prev, step3prev, step3, step42ns, step52ns, verifier_code, verifier_query_code; };

// ---- starkinfo_codegen.rs:421-578 ---------------------------------------------------------------------------------
struct ContextC { i64 exp_id; i64 tmp_used; std::vector<Section> code; };

static Node eval_single_op(ContextC& cc, const Expr& exp, bool prime, std::vector<Node>& values) {      // :442-555
    const std::string& op = exp.op;
    auto tmp = [&] { Node r = N("tmp", cc.tmp_used); cc.tmp_used += 1; return r; };
    if (op == "add" || op == "sub" || op == "mul" || op == "muladd") { Node r = tmp(); cc.code.push_back({op, r, values}); return r; }
    if (op == "addc" || op == "mulc") {
        Node r = tmp(); cc.code.push_back({op == "addc" ? "add" : "mul", r, {values[0], N_num(std::to_string(exp.const_))}}); return r;
    }
    if (op == "neg") { Node r = tmp(); cc.code.push_back({"sub", r, {N_num("0"), values[0]}}); return r; }
    if (op == "cm" || op == "const" || op == "exp" || op == "q") {
        if (exp.next && prime) throw Error("Double Prime");
        return N(op, exp.id, exp.next || prime);
    }
    if (op == "number") return N_num(exp.value);
    if (op == "public" || op == "challenge" || op == "eval") return N(op, exp.id);
    if (op == "xDivXSubXi" || op == "xDivXSubWXi" || op == "x") return N(op, 0);
    throw Error("InvalidOperator: eval_exp: " + op);
}
static Node eval_exp(ContextC& cc, const Expr& exp, bool prime) {                                       // :421-440, left-to-right post-order
    if (is_nop(exp)) throw Error("nop expression");
    std::vector<Node> vals;
    if (exp.has_values) for (auto& v : exp.values) vals.push_back(eval_exp(cc, v, prime));
    return eval_single_op(cc, exp, prime, vals);
}
static Expr find_muladd(const Expr& exp) {                                                              // :358-384
    if (exp.has_values) {
        const auto& v = exp.values;
        if (exp.op == "add" && v[0].op == "mul") {
            Expr r = E("muladd"); r.has_values = true;
            r.values = {find_muladd(v[0].values[0]), find_muladd(v[0].values[1]), find_muladd(v[1])};
            return r;
        }
        if (exp.op == "add" && v[1].op == "mul") {
            Expr r = E("muladd"); r.has_values = true;
            r.values = {find_muladd(v[1].values[0]), find_muladd(v[1].values[1]), find_muladd(v[0])};
            return r;
        }
        Expr r = exp;
        if (!v.empty()) { r.values.clear(); for (auto& x : v) r.values.push_back(find_muladd(x)); }
        return r;
    }
    return exp;
}
static void pil_code_gen(Ctx& ctx, Pil& pil, i64 exp_id, bool prime, const std::string& res_type, i64 res_id, bool muladd);
static void calculate_deps(Ctx& ctx, Pil& pil, const Expr& expr, bool prime, i64 exp_id, bool muladd) { // :557-578
    if (expr.op == "exp") {
        if (prime && expr.next) throw Error("Double prime");
        pil_code_gen(ctx, pil, expr.id, prime || expr.next, "", 0, muladd);
    }
    if (expr.has_values) for (auto& e : expr.values) calculate_deps(ctx, pil, e, prime, exp_id, muladd);
}
static void pil_code_gen(Ctx& ctx, Pil& pil, i64 exp_id, bool prime, const std::string& res_type, i64 res_id, bool muladd) {   // :296-356
    const auto key = std::make_pair(prime ? 1 : 0, exp_id);
    if (ctx.calculated.count(key)) {                                       // contains_key, whatever the value
        if (!res_type.empty()) {
            for (auto& c : ctx.code)
                if (c.exp_id == exp_id && c.prime == prime) { c.code.push_back({"copy", N(res_type, res_id, prime), {c.code.back().dest}}); break; }
        }
        return;
    }
    ZK_REQUIRE(exp_id >= 0 && (size_t)exp_id < pil.expressions.size(), "expression id out of range");
    { const Expr exp = pil.expressions[exp_id]; calculate_deps(ctx, pil, exp, prime, exp_id, false); }
    ContextC cc{exp_id, ctx.tmp_used, {}};
    const Expr exp = muladd ? find_muladd(pil.expressions[exp_id]) : pil.expressions[exp_id];
    Node ret = eval_exp(cc, exp, prime);
    if (ret.type_ == "tmp") { cc.code.back().dest = N("exp", exp_id, prime); cc.tmp_used -= 1; }
    else cc.code.push_back({"copy", N("exp", exp_id, prime), {ret}});
    if (!res_type.empty()) {
        if (prime) throw Error("Prime in retType");
        cc.code.push_back({"copy", N(res_type, res_id, prime), {N("exp", exp_id, prime)}});
    }
    ctx.code.push_back({exp_id, prime, cc.code});
    ctx.calculated[key] = true;
    if (cc.tmp_used > ctx.tmp_used) ctx.tmp_used = cc.tmp_used;
}
static std::vector<Section> build_linear_code(const Ctx& ctx, const Pil& pil, const char* loop_pos) {   // :610-658
    std::map<i64, int> calc;                                               // NB keyed by expression id, looked up by code index (as the reference does)
    const bool filtered = strcmp(loop_pos, "first") != 0;
    if (filtered)
        for (auto& c : ctx.code) {
            const Expr& e = pil.expressions[c.exp_id];
            if (e.has_idq || e.keep || e.keep2ns) calc[c.exp_id] |= c.prime ? 2 : 1;
        }
    std::vector<Section> res;
    for (size_t i = 0; i < ctx.code.size(); ++i) {
        const auto& c = ctx.code[i];
        auto it = calc.find((i64)i);
        const bool both = filtered && it != calc.end() && it->second == 3;
        if (both && ((strcmp(loop_pos, "i") == 0 && !c.prime) || strcmp(loop_pos, "last") == 0)) continue;
        res.insert(res.end(), c.code.begin(), c.code.end());
    }
    return res;
}
static Segment build_code(Ctx& ctx, const Pil& pil) {                                                   // :586-603
    Segment seg;
    seg.first = build_linear_code(ctx, pil, "first"); seg.i = build_linear_code(ctx, pil, "i"); seg.last = build_linear_code(ctx, pil, "last");
    seg.tmp_used = ctx.tmp_used;
    for (size_t i = 0; i < pil.expressions.size(); ++i) {
        const Expr& e = pil.expressions[i];
        if (!e.keep && !e.has_idq) { ctx.calculated[{0, (i64)i}] = false; ctx.calculated[{1, (i64)i}] = false; }
    }
    ctx.code.clear();
    return seg;
}
template <class F>
static void iterate_code(Segment& seg, F&& f) {                                                         // :660-669
    for (auto* part : {&seg.first, &seg.i, &seg.last})
        for (auto& c : *part) { for (auto& s : c.src) f(s); f(c.dest); }
}

// ---- degree (starkinfo_cp_prover.rs:243-270) and dimension (starkinfo_map.rs:619-640) ------------------------------
static i64 exp_degree(const Pil& pil, const Expr& exp) {
    const std::string& op = exp.op;
    if (op == "add" || op == "sub" || op == "addc" || op == "mulc" || op == "neg") {
        i64 m = 1; for (auto& x : exp.values) m = std::max(m, exp_degree(pil, x)); return m;
    }
    if (op == "mul") return exp_degree(pil, exp.values[0]) + exp_degree(pil, exp.values[1]);
    if (op == "muladd") return std::max(exp_degree(pil, exp.values[0]) + exp_degree(pil, exp.values[1]), exp_degree(pil, exp.values[2]));
    if (op == "cm" || op == "const" || op == "x") return 1;
    if (op == "exp") return exp_degree(pil, pil.expressions[exp.id]);
    if (op == "number" || op == "public" || op == "challenge" || op == "eval") return 0;
    throw Error("Exp op not defined: " + op);
}
static i64 exp_dim(const Pil& pil, const Expr& exp) {
    const std::string& op = exp.op;
    if (op == "add" || op == "sub" || op == "mul" || op == "muladd" || op == "addc" || op == "mulc" || op == "neg") {
        i64 m = 1; for (auto& x : exp.values) m = std::max(m, exp_dim(pil, x)); return m;
    }
    if (op == "cm") return pil.cm_dims.at(exp.id);
    if (op == "exp") return exp_dim(pil, pil.expressions[exp.id]);
    if (op == "q") return exp_dim(pil, pil.expressions[pil.q2exp.at(exp.id)]);
    if (op == "const" || op == "number" || op == "public" || op == "x") return 1;
    if (op == "challenge" || op == "eval" || op == "xDivXSubXi" || op == "xDivXSubWXi") return 3;
    throw Error("Exp op not defined: " + op);
}

// ---- calculate_im_pols (starkinfo_cp_prover.rs:123-291) ------------------------------------------------------------
typedef std::map<i64, bool> ImMap;
struct ImRes { bool some; ImMap im; i64 deg; };
static bool is_scalar_op(const std::string& op) { return op == "number" || op == "public" || op == "challenge"; }
static ImRes calc_im(const Pil& pil, const Expr& exp, bool have, const ImMap& im, i64 max_deg, i64 abs_max, i64& st) {
    if (!have) return {false, {}, -1};
    const std::string& op = exp.op;
    if (op == "add" || op == "sub" || op == "addc" || op == "mulc" || op == "neg") {
        i64 md = 0; ImRes cur{true, im, 0};
        for (auto& v : exp.values) { cur = calc_im(pil, v, cur.some, cur.im, max_deg, abs_max, st); md = std::max(md, cur.deg); }
        return {cur.some, cur.im, md};
    }
    if (is_scalar_op(op)) return {true, im, 0};
    if (op == "x" || op == "const" || op == "cm") return max_deg < 1 ? ImRes{false, {}, -1} : ImRes{true, im, 1};
    if (op == "mul") {
        const auto& v = exp.values;
        if (is_scalar_op(v[0].op)) return calc_im(pil, v[1], true, im, max_deg, abs_max, st);
        if (is_scalar_op(v[1].op)) return calc_im(pil, v[0], true, im, max_deg, abs_max, st);
        const i64 here = exp_degree(pil, exp);
        if (here <= max_deg) return {true, im, here};
        ImRes best{false, {}, -1};
        for (i64 l = 0; l <= max_deg; ++l) {
            ImRes r1 = calc_im(pil, v[0], true, im, l, abs_max, st);
            ImRes r2 = calc_im(pil, v[1], r1.some, r1.im, max_deg - l, abs_max, st);
            if (r2.some && (!best.some || r2.im.size() < best.im.size())) best = {true, r2.im, r1.deg + r2.deg};
            if (best.some && best.im.size() == im.size()) return best;
        }
        return best;
    }
    if (op == "exp") {
        if (max_deg < 1) return {false, {}, -1};
        if (im.count(exp.id)) return {true, im, 1};
        ImRes r = calc_im(pil, pil.expressions[exp.id], true, im, abs_max, abs_max, st);
        if (!r.some) return {false, {}, -1};
        if (r.deg > max_deg) { r.im[exp.id] = true; st = std::max(st, r.deg); return {true, r.im, 1}; }
        return r;
    }
    throw Error("Exp op not defined: " + op);
}

static unsigned long long mulp(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) % GL_P); }

// ---- StarkInfo::new + map -------------------------------------------------------------------------------------------
struct Gen {
    Pil pil; Info info; Program prog;
    i64 nbits = 0, nbits_ext = 0;
    std::string l1_name = GLOBAL_L1;

    Expr lc(const std::vector<i64>& ids, const Expr& u, bool left_mul) {       // Horner in u over expressions `ids`
        Expr acc = E("nop");
        for (i64 j : ids) { Expr e = e_exp(j); acc = is_nop(acc) ? e : e_add(left_mul ? e_mul(u, acc) : e_mul(acc, u), e); }
        return acc;
    }
    i64 push_expr(const Expr& e) { pil.expressions.push_back(e); return (i64)pil.expressions.size() - 1; }
    Expr l1_const() {
        auto it = pil.references.find(l1_name);
        if (it == pil.references.end()) throw Error(l1_name + " must be defined");
        return e_const(it->second.first);
    }
    i64 push_identity(Expr e) { e.deg = 2; const i64 i = push_expr(e); pil.pol_identities.push_back(i); return i; }

    void load(const JVal& P, const JVal& S) {
        pil.nCommitments = P.at("nCommitments").i64(); pil.nQ = P.at("nQ").i64(); pil.nConstants = P.at("nConstants").i64();
        for (auto& p : P.at("publics").arr) {
            Public q; q.polType = p.at("polType").str(); q.polId = p.at("polId").i64(); q.idx = p.at("idx").i64(); q.id = p.at("id").i64();
            q.name = p.at("name").str(); pil.publics.push_back(q);
        }
        for (auto& kv : P.at("references").obj) {
            pil.references[kv.first] = {kv.second.at("id").i64(), kv.second.at("polDeg").i64()};
            if (pil.first_pol_deg < 0) pil.first_pol_deg = kv.second.at("polDeg").i64();
        }
        for (auto& e : P.at("expressions").arr) pil.expressions.push_back(load_expr(e));
        for (auto& e : P.at("polIdentities").arr) pil.pol_identities.push_back(e.at("e").i64());
        auto ids = [](const JVal& v) { std::vector<i64> o; for (auto& x : v.arr) o.push_back(x.i64()); return o; };
        auto idents = [&](const char* key, std::vector<Identity>& out) {
            const JVal* L = P.find(key);
            if (!L || L->is_null()) return;
            for (auto& d : L->arr) {
                Identity x; x.f = ids(d.at("f")); x.t = ids(d.at("t"));
                if (const JVal* v = d.find("selF")) if (!v->is_null()) { x.has_self = true; x.self_ = v->i64(); }
                if (const JVal* v = d.find("selT")) if (!v->is_null()) { x.has_selt = true; x.selt = v->i64(); }
                out.push_back(x);
            }
        };
        idents("plookupIdentities", pil.plookups); idents("permutationIdentities", pil.permutations);
        if (const JVal* L = P.find("connectionIdentities")) if (!L->is_null())
            for (auto& d : L->arr) pil.connections.push_back({ids(d.at("pols")), ids(d.at("connections"))});
        nbits = S.at("nBits").i64(); nbits_ext = S.at("nBitsExt").i64();
        ZK_REQUIRE(!pil.references.empty(), "pil: no references");
        if (((i64)1 << nbits) != pil.first_pol_deg) throw Error("stark_deg != pil_deg");
        if (nbits_ext != S.at("steps").at(0).at("nBits").i64()) throw Error("MustEqualDegreeError: stark_struct.nBitsExt != stark_struct.steps[0].nBits");
        for (const char* s : SECTIONS) { info.map_sections[s]; info.map_sectionsN1[s] = 0; info.map_sectionsN3[s] = 0; info.map_sectionsN[s] = 0; info.map_offsets[s] = 0; info.map_deg[s] = 0; }
        info.n_constants = pil.nConstants; info.n_publics = (i64)pil.publics.size();
    }

    void run() {
        // -- generate_public_calculators (starkinfo.rs:274-322)
        for (auto& p : pil.publics) {
            if (p.polType != "imP") continue;
            Ctx ctx;
            pil_code_gen(ctx, pil, p.polId, false, "", 0, false);
            Segment seg = build_code(ctx, pil);
            std::map<std::pair<int, i64>, i64> m; i64 tmp_used = seg.tmp_used;
            iterate_code(seg, [&](Node& r) {
                if (r.type_ != "exp") return;
                auto k = std::make_pair(r.prime ? 1 : 0, r.id);
                if (!m.count(k)) m[k] = tmp_used++;
                r.prime = false; r.type_ = "tmp"; r.id = m[k];
            });
            seg.tmp_used = tmp_used;
            prog.publics_code.push_back(seg);
        }
        info.n_cm1 = pil.nCommitments;
        Ctx ctx, ctx2ns;

        // -- generate_step2 (starkinfo.rs:324-408): plookup h1, h2
        const Expr u = e_challenge(CH_U), def_val = e_challenge(CH_DEFVAL);
        for (auto& pi : pil.plookups) {
            Expr t_exp = lc(pi.t, u, true);
            if (pi.has_selt) { t_exp = e_add(e_mul(e_sub(t_exp, def_val), e_exp(pi.selt)), def_val); t_exp.has_idq = true; t_exp.idq = pil.nQ++; }
            t_exp.keep = true; const i64 t_id = push_expr(t_exp);
            Expr f_exp = lc(pi.f, u, false);
            if (pi.has_self) { f_exp = e_add(e_mul(e_sub(f_exp, e_exp(t_id)), e_exp(pi.self_)), e_exp(t_id)); f_exp.has_idq = true; f_exp.idq = pil.nQ++; }
            f_exp.keep = true; const i64 f_id = push_expr(f_exp);
            pil_code_gen(ctx, pil, f_id, false, "", 0, false);
            pil_code_gen(ctx, pil, t_id, false, "", 0, false);
            PCCtx c; c.f_exp_id = f_id; c.t_exp_id = t_id; c.h1_id = pil.nCommitments; c.h2_id = pil.nCommitments + 1;
            pil.nCommitments += 2;
            info.pu_ctx.push_back(c);
        }
        prog.step2prev = build_code(ctx, pil);
        ctx.calculated.clear();
        info.n_cm2 = pil.nCommitments - info.n_cm1;

        // -- generate_step3 (starkinfo_Z.rs)
        const Expr gamma = e_challenge(CH_GAMMA), beta = e_challenge(CH_BETA), one = e_number("1");
        for (auto& pi : pil.permutations) {                                  // generate_permutation_LC :32-105
            Expr t_exp = lc(pi.t, u, true);
            if (pi.has_selt) { t_exp = e_add(e_mul(e_sub(t_exp, def_val), e_exp(pi.selt)), def_val); t_exp.has_idq = true; t_exp.idq = pil.nQ++; }
            const i64 t_id = push_expr(t_exp);
            Expr f_exp = lc(pi.f, u, false);
            if (pi.has_self) { f_exp = e_add(e_mul(e_sub(f_exp, def_val), e_exp(pi.self_)), def_val); f_exp.has_idq = true; f_exp.idq = pil.nQ++; }
            const i64 f_id = push_expr(f_exp);
            PCCtx c; c.f_exp_id = f_id; c.t_exp_id = t_id; info.pe_ctx.push_back(c);
        }
        for (size_t i = 0; i < pil.plookups.size(); ++i) {                   // generate_plookup_Z :108-199
            PCCtx& pu = info.pu_ctx[i];
            pu.z_id = pil.nCommitments++;
            const Expr h1 = e_cm(pu.h1_id), h2 = e_cm(pu.h2_id), h1p = e_cm(pu.h1_id, true);
            const Expr f = e_exp(pu.f_exp_id), t = e_exp(pu.t_exp_id), tp = e_exp(pu.t_exp_id, true);
            const Expr z = e_cm(pu.z_id), zp = e_cm(pu.z_id, true);
            pu.c1_id = push_identity(e_mul(l1_const(), e_sub(z, one)));
            const Expr g1b = e_mul(gamma, e_add(one, beta));
            Expr num = e_mul(e_mul(e_add(f, gamma), e_add(e_add(t, e_mul(tp, beta)), g1b)), e_add(one, beta));
            num.has_idq = true; num.idq = pil.nQ++; num.keep = true;
            pu.num_id = push_expr(num);
            Expr den = e_mul(e_add(e_add(h1, e_mul(h2, beta)), g1b), e_add(e_add(h2, e_mul(h1p, beta)), g1b));
            den.has_idq = true; den.idq = pil.nQ++; den.keep = true;
            pu.den_id = push_expr(den);
            pu.c2_id = push_identity(e_sub(e_mul(zp, e_exp(pu.den_id)), e_mul(z, e_exp(pu.num_id))));
            pil_code_gen(ctx, pil, pu.num_id, false, "", 0, false);
            pil_code_gen(ctx, pil, pu.den_id, false, "", 0, false);
        }
        for (size_t i = 0; i < pil.permutations.size(); ++i) {               // generate_permutation_Z :201-271
            PCCtx& pe = info.pe_ctx[i];
            pe.z_id = pil.nCommitments++;
            const Expr f = e_exp(pe.f_exp_id), t = e_exp(pe.t_exp_id), z = e_cm(pe.z_id), zp = e_cm(pe.z_id, true);
            pe.c1_id = push_identity(e_mul(l1_const(), e_sub(z, one)));
            Expr num = e_add(f, beta); num.keep = true; pe.num_id = push_expr(num);
            Expr den = e_add(t, beta); den.keep = true; pe.den_id = push_expr(den);
            pe.c2_id = push_identity(e_sub(e_mul(zp, e_exp(pe.den_id)), e_mul(z, e_exp(pe.num_id))));
            pil_code_gen(ctx, pil, pe.num_id, false, "", 0, false);
            pil_code_gen(ctx, pil, pe.den_id, false, "", 0, false);
        }
        for (auto& ci : pil.connections) {                                   // generate_connections_Z :273-423
            PCCtx c; c.z_id = pil.nCommitments++;
            Expr num = e_add(e_add(e_exp(ci.pols[0]), e_mul(beta, E("x"))), gamma); num.keep = true;
            Expr den = e_add(e_add(e_exp(ci.pols[0]), e_mul(beta, e_exp(ci.connections[0]))), gamma); den.keep = true;
            c.num_id = push_expr(num); c.den_id = push_expr(den);
            unsigned long long k = 12275445934081160404ULL, kc = k;          // helper.rs:16-23 get_ks
            for (size_t i = 1; i < ci.pols.size(); ++i) {
                num = e_mul(e_exp(c.num_id), e_add(e_add(e_exp(ci.pols[i]), e_mul(e_mul(beta, e_number(std::to_string(kc))), E("x"))), gamma));
                num.has_idq = true; num.idq = pil.nQ++;
                den = e_mul(e_exp(c.den_id), e_add(e_add(e_exp(ci.pols[i]), e_mul(beta, e_exp(ci.connections[i]))), gamma));
                den.has_idq = true; den.idq = pil.nQ++;
                c.num_id = push_expr(num); c.den_id = push_expr(den);
                kc = mulp(kc, k);
            }
            const Expr z = e_cm(c.z_id), zp = e_cm(c.z_id, true);
            c.c1_id = push_identity(e_mul(l1_const(), e_sub(z, one)));
            c.c2_id = push_identity(e_sub(e_mul(zp, e_exp(c.den_id)), e_mul(z, e_exp(c.num_id))));
            pil_code_gen(ctx, pil, c.num_id, false, "", 0, false);
            pil_code_gen(ctx, pil, c.den_id, false, "", 0, false);
            info.ci_ctx.push_back(c);
        }
        prog.step3prev = build_code(ctx, pil);
        ctx.calculated.clear();

        // -- generate_constraint_polynomial (starkinfo_cp_prover.rs:12-119)
        const Expr vc = e_challenge(CH_VC);
        Expr c_exp = E("nop");
        for (i64 id : pil.pol_identities) { Expr e = e_exp(id); c_exp = is_nop(c_exp) ? e : e_add(e_mul(vc, c_exp), e); }
        const i64 max_deg = ((i64)1 << (nbits_ext - nbits)) + 1;
        for (i64 d = 2; d <= max_deg; ++d) {
            i64 st = 0;
            ImRes r = calc_im(pil, c_exp, true, {}, d, d, st);
            const i64 qd = std::max(r.deg, st) - 1;
            if (r.some && (info.q_deg == 0 || (i64)r.im.size() + qd < (i64)info.im_exps.size() + info.q_deg)) { info.q_deg = qd; info.im_exps = r.im; }
        }
        for (auto& kv : info.im_exps) info.im_exps_list.push_back(kv.first);  // sorted (std::map)
        for (i64 k : info.im_exps_list) {
            info.im_exp2cm[k] = pil.nCommitments++;
            Expr e = E2("sub", pil.expressions[k], E_id("cm", pil.nCommitments - 1));
            c_exp = is_nop(c_exp) ? e : e_add(e_mul(vc, c_exp), e);
        }
        info.c_exp = push_expr(c_exp);
        info.n_cm3 = pil.nCommitments - info.n_cm1 - info.n_cm2;
        for (i64 i = 0; i < info.q_deg; ++i) info.qs.push_back(pil.nCommitments++);
        for (i64 k : info.im_exps_list) pil_code_gen(ctx, pil, k, false, "", 0, false);
        prog.step3 = build_code(ctx, pil);
        for (auto& kv : info.im_exps) { ctx2ns.calculated[{0, kv.first}] = kv.second; ctx2ns.calculated[{1, kv.first}] = kv.second; }
        pil_code_gen(ctx2ns, pil, info.c_exp, false, "", 0, false);
        { auto& code = ctx2ns.code.back().code; code.push_back({"mul", N("q", 0), {code.back().dest, N("Zi", 0)}}); }
        prog.step42ns = build_code(ctx2ns, pil);
        info.n_cm4 = info.q_deg;

        // -- generate_constraint_polynomial_verifier (starkinfo_cp_ver.rs:8-119)
        {
            Ctx cv;
            for (auto& kv : info.im_exps) { cv.calculated[{0, kv.first}] = kv.second; cv.calculated[{1, kv.first}] = kv.second; }
            pil_code_gen(cv, pil, info.c_exp, false, "", 0, true);
            Segment code = build_code(cv, pil);
            std::map<std::pair<int, i64>, i64> m; i64 tmp_used = code.tmp_used;
            std::set<i64> im_set(info.im_exps_list.begin(), info.im_exps_list.end());
            auto ev_index = [&](const std::string& type_, int p, i64 id, bool prime) {
                auto& list = type_ == "cm" ? info.ev_cm : info.ev_const;
                for (auto& kv : list) if (kv.first.first == p && kv.first.second == id) return kv.second;
                const i64 idx = (i64)info.ev_map.size();
                list.push_back({{p, id}, idx});
                info.ev_map.push_back(N(type_, id, prime));
                return idx;
            };
            iterate_code(code, [&](Node& r) {
                const int p = r.prime ? 1 : 0;
                const std::string t = r.type_;
                if (t == "exp") {
                    if (im_set.count(r.id)) {
                        r.type_ = "cm"; r.id = info.im_exp2cm[r.id];
                        const i64 idx = ev_index("cm", p, r.id, r.prime);
                        r.prime = false; r.id = idx; r.type_ = "eval";
                    } else {
                        auto k = std::make_pair(p, r.id);
                        if (!m.count(k)) m[k] = tmp_used++;
                        r.type_ = "tmp"; r.exp_id = r.id; r.id = m[k];
                    }
                } else if (t == "cm" || t == "const") {
                    const i64 idx = ev_index(t, p, r.id, r.prime);
                    r.prime = false; r.id = idx; r.type_ = "eval";
                } else if (!(t == "number" || t == "challenge" || t == "public" || t == "tmp" || t == "Z" || t == "x" || t == "eval"))
                    throw Error("Invalid reference type: " + t);
            });
            for (i64 i = 0; i < info.q_deg; ++i) {
                info.ev_cm.push_back({{0, info.qs[i]}, (i64)info.ev_map.size()});
                info.ev_map.push_back(N("cm", info.qs[i]));
            }
            code.tmp_used = tmp_used;
            prog.verifier_code = code;
        }

        // -- generate_fri_polynomial (starkinfo_fri_prover.rs:10-98), with the prover's ctx2ns
        const Expr vf1 = e_challenge(CH_VF1), vf2 = e_challenge(CH_VF2);
        Expr fri = E("nop");
        for (i64 i = 0; i < pil.nCommitments; ++i) fri = is_nop(fri) ? e_cm(i) : e_add(e_mul(vf1, fri), e_cm(i));
        Expr fri1 = E("nop"), fri2 = E("nop");
        for (size_t i = 0; i < info.ev_map.size(); ++i) {
            const Node& ev = info.ev_map[i];
            Expr& cur = ev.prime ? fri2 : fri1;
            Expr e = ev.type_ == "cm" ? e_cm(ev.id) : ev.type_ == "q" ? e_q(ev.id) : ev.type_ == "const" ? e_const(ev.id) : throw Error("ev_map: bad type");
            cur = is_nop(cur) ? e_sub(e, e_eval((i64)i)) : e_add(e_mul(cur, vf2), e_sub(e, e_eval((i64)i)));
        }
        if (!is_nop(fri)) {                                                  // sic: tests fri_exp (fri_prover.rs:64)
            fri1 = e_mul(fri1, E("xDivXSubXi"));
            fri = !is_nop(fri) ? e_add(e_mul(vf1, fri), fri1) : fri1;
        }
        if (!is_nop(fri2)) {
            fri2 = e_mul(fri2, E("xDivXSubWXi"));
            fri = !is_nop(fri) ? e_add(e_mul(vf1, fri), fri2) : fri2;
        }
        fri.keep2ns = true;
        info.fri_exp_id = push_expr(fri);
        pil_code_gen(ctx2ns, pil, info.fri_exp_id, false, "f", 0, false);
        ctx2ns.code.back().code.back().dest = N("f", 0);
        prog.step52ns = build_code(ctx2ns, pil);

        // -- generate_fri_verifier (starkinfo_fri_ver.rs:7-21)
        { Ctx cq; pil_code_gen(cq, pil, info.fri_exp_id, false, "", 0, true); prog.verifier_query_code = build_code(cq, pil); }
        info.n_exps = (i64)pil.expressions.size();
        map();
    }

    // ---- StarkInfo::map (starkinfo_map.rs:10-307) ---------------------------------------------------------------------
    std::map<i64, i64> tmpexps;
    i64 add_pol(const std::string& section, i64 dim) { info.var_pol_map.push_back({section, 0, dim, 0}); return (i64)info.var_pol_map.size() - 1; }
    i64 add_cm(const std::string& sec, i64 dim) {
        const i64 pn = add_pol(sec + "_n", dim), p2 = add_pol(sec + "_2ns", dim);
        info.cm_n.push_back(pn); info.cm_2ns.push_back(p2);
        info.map_sections[sec + "_n"].push_back(pn); info.map_sections[sec + "_2ns"].push_back(p2);
        return pn;
    }
    void add_tmpexp(i64 exp_id, i64 dim) {
        auto it = info.im_exps.find(exp_id);
        const bool im_none = it == info.im_exps.end() || !it->second;
        if (im_none && !tmpexps.count(exp_id)) {
            tmpexps[exp_id] = (i64)info.tmpexp_n.size();
            const i64 pp = add_pol("tmpexp_n", dim);
            info.tmpexp_n.push_back(pp); info.map_sections["tmpexp_n"].push_back(pp); info.exp2pol[exp_id] = pp;
        }
    }
    void fix_prover_code(Segment& seg, bool dom_n) {                           // :427-488
        std::map<std::pair<int, i64>, i64> m; i64 tmp_used = seg.tmp_used;
        std::set<i64> im_set(info.im_exps_list.begin(), info.im_exps_list.end());
        iterate_code(seg, [&](Node& r) {
            const std::string t = r.type_;
            if (t == "cm") r.p = dom_n ? info.cm_n.at(r.id) : info.cm_2ns.at(r.id);
            else if (t == "exp") {
                if (im_set.count(r.id)) { r.type_ = "cm"; r.id = info.im_exp2cm[r.id]; }
                else if (tmpexps.count(r.id) && dom_n) { r.type_ = "tmpExp"; r.dim = exp_dim(pil, pil.expressions[r.id]); r.id = tmpexps[r.id]; }
                else {
                    auto k = std::make_pair(r.prime ? 1 : 0, r.id);
                    if (!m.count(k)) m[k] = tmp_used++;
                    r.type_ = "tmp"; r.exp_id = r.id; r.id = m[k];
                }
            } else if (!(t == "const" || t == "number" || t == "challenge" || t == "public" || t == "tmp" || t == "Zi" || t == "xDivXSubXi" ||
                         t == "xDivXSubWXi" || t == "eval" || t == "x" || t == "q" || t == "f" || t == "tmpExp"))
                throw Error("Invalid reference type " + t);
        });
        seg.tmp_used = tmp_used;
    }
    void set_code_dimensions(Segment& seg, i64 dim_x) {                        // :309-425
        std::map<i64, i64> tmp_dim;
        auto get_dim = [&](Node& r) {
            const std::string& t = r.type_;
            i64 d;
            if (t == "tmp") { auto it = tmp_dim.find(r.id); if (it == tmp_dim.end()) throw Error("tmp used before set"); d = it->second; }
            else if (t == "tree1" || t == "tree2" || t == "tree3" || t == "tree4" || t == "tmpExp") d = r.dim;
            else if (t == "cm") d = info.var_pol_map.at(info.cm_2ns.at(r.id)).dim;
            else if (t == "q") d = info.var_pol_map.at(info.qs.at(r.id)).dim;
            else if (t == "const" || t == "number" || t == "public" || t == "Zi") d = 1;
            else if (t == "eval" || t == "challenge" || t == "Z") d = 3;
            else if (t == "xDivXSubXi" || t == "xDivXSubWXi" || t == "x") d = dim_x;
            else throw Error("Invalid reference type get " + t);
            if (d == 0) throw Error("Invalid dim");
            r.dim = d;
            return d;
        };
        for (auto* part : {&seg.first, &seg.i, &seg.last})
            for (auto& c : *part) {
                i64 nd;
                if (c.op == "add" || c.op == "sub" || c.op == "mul") nd = std::max(get_dim(c.src[0]), get_dim(c.src[1]));
                else if (c.op == "muladd") nd = std::max(std::max(get_dim(c.src[0]), get_dim(c.src[1])), get_dim(c.src[2]));
                else if (c.op == "copy") nd = get_dim(c.src[0]);
                else throw Error("Invalid op: " + c.op);
                Node& d = c.dest;
                if (d.type_ == "tmp") { tmp_dim[d.id] = nd; d.dim = nd; }
                else if (d.type_ == "exp" || d.type_ == "cm" || d.type_ == "q" || d.type_ == "tmpExp" || d.type_ == "f") d.dim = nd;
                else throw Error("Invalid reference type set " + d.type_);
            }
    }
    void map() {
        const i64 n1 = info.n_cm1, n2 = info.n_cm2, n3 = info.n_cm3, n4 = info.n_cm4;
        pil.cm_dims.assign((size_t)(n1 + n2 + n3 + n4), 0);
        for (i64 i = 0; i < n1; ++i) { add_cm("cm1", 1); pil.cm_dims[i] = 1; }
        for (size_t i = 0; i < info.pu_ctx.size(); ++i) {
            const PCCtx& pu = info.pu_ctx[i];
            const i64 dim = std::max(exp_dim(pil, pil.expressions[pu.f_exp_id]), exp_dim(pil, pil.expressions[pu.t_exp_id]));
            add_cm("cm2", dim); pil.cm_dims[n1 + i * 2] = dim;
            add_cm("cm2", dim); pil.cm_dims[n1 + i * 2 + 1] = dim;
            add_tmpexp(pu.f_exp_id, dim); add_tmpexp(pu.t_exp_id, dim);
        }
        {
            std::vector<PCCtx> all = info.pu_ctx; all.insert(all.end(), info.pe_ctx.begin(), info.pe_ctx.end()); all.insert(all.end(), info.ci_ctx.begin(), info.ci_ctx.end());
            for (size_t i = 0; i < all.size(); ++i) { add_cm("cm3", 3); pil.cm_dims[n1 + n2 + i] = 3; add_tmpexp(all[i].num_id, 3); add_tmpexp(all[i].den_id, 3); }
        }
        for (size_t i = 0; i < info.im_exps_list.size(); ++i) {
            const i64 k = info.im_exps_list[i];
            const i64 dim = exp_dim(pil, pil.expressions[k]);
            const i64 pn = add_cm("cm3", dim);
            pil.cm_dims[n1 + n2 + i] = dim;                                   // sic (starkinfo_map.rs:186)
            info.exp2pol[k] = pn;
        }
        info.q_dim = exp_dim(pil, pil.expressions[info.c_exp]);
        for (i64 i = 0; i < info.q_deg; ++i) { add_cm("cm4", info.q_dim); pil.cm_dims[n1 + n2 + n3 + i] = info.q_dim; }
        info.q_2ns.push_back(add_pol("q_2ns", info.q_dim));
        info.f_2ns.push_back(add_pol("f_2ns", 3));
        for (const char* s : SECTIONS) {                                     // map_section :490-516
            i64 p = 0;
            for (i64 e = 1; e <= 3; ++e) {
                for (auto& pp : info.var_pol_map) if (pp.section == s && pp.dim == e) { pp.section_pos = p; p += e; }
                if (e == 1) info.map_sectionsN1[s] = p;
                if (e == 3) info.map_sectionsN[s] = p;
            }
            info.map_sectionsN3[s] = (info.map_sectionsN[s] - info.map_sectionsN1[s]) / 3;
        }
        const i64 N = (i64)1 << nbits, Next = (i64)1 << nbits_ext;
        const std::pair<const char*, i64> order[11] = {{"cm1_n", N}, {"cm2_n", N}, {"cm3_n", N}, {"cm4_n", N}, {"tmpexp_n", N}, {"cm1_2ns", Next},
                                                         {"cm2_2ns", Next}, {"cm3_2ns", Next}, {"cm4_2ns", Next}, {"q_2ns", Next}, {"f_2ns", Next}};
        i64 acc = 0;
        for (auto& o : order) { info.map_offsets[o.first] = acc; acc += o.second * info.map_sectionsN[o.first]; info.map_deg[o.first] = o.second; }
        info.map_total_n = acc;

        for (auto& seg : prog.publics_code) fix_prover_code(seg, true);
        fix_prover_code(prog.step2prev, true); fix_prover_code(prog.step3prev, true); fix_prover_code(prog.step3, true);
        fix_prover_code(prog.step42ns, false); fix_prover_code(prog.step52ns, false); fix_prover_code(prog.verifier_query_code, false);
        iterate_code(prog.verifier_query_code, [&](Node& r) {                 // :257-283
            if (r.type_ != "cm") return;
            const PolInfo& p1 = info.var_pol_map.at(info.cm_2ns.at(r.id));
            r.type_ = p1.section == "cm1_2ns" ? "tree1" : p1.section == "cm2_2ns" ? "tree2" : p1.section == "cm3_2ns" ? "tree3" : "tree4";
            r.tree_pos = p1.section_pos; r.dim = p1.dim;
        });
        for (i64 i = 0; i < info.n_publics; ++i)
            if ((size_t)i < prog.publics_code.size()) {
                Segment& s = prog.publics_code[i];
                if (!s.first.empty() || !s.i.empty() || !s.last.empty()) set_code_dimensions(s, 1);
            }
        set_code_dimensions(prog.step2prev, 1); set_code_dimensions(prog.step3prev, 1); set_code_dimensions(prog.step3, 1);
        set_code_dimensions(prog.step42ns, 1); set_code_dimensions(prog.step52ns, 1);
        set_code_dimensions(prog.verifier_code, 3); set_code_dimensions(prog.verifier_query_code, 1);
    }

    // ---- serde_json shape (HashMap<usize, _> keys as strings, EVIdx maps as [[p, id], idx] lists) ----------------------
    static void jstr(std::ostringstream& o, const std::string& s) {
        o << '"';
        for (char c : s) { if (c == '"' || c == '\\') o << '\\'; o << c; }
        o << '"';
    }
    static void jnode(std::ostringstream& o, const Node& n) {
        o << "{\"type_\":"; jstr(o, n.type_); o << ",\"id\":" << n.id << ",\"value\":";
        if (n.has_value) jstr(o, n.value); else o << "null";
        o << ",\"dim\":" << n.dim << ",\"prime\":" << (n.prime ? "true" : "false") << ",\"tree_pos\":" << n.tree_pos << ",\"p\":" << n.p
          << ",\"exp_id\":" << n.exp_id << "}";
    }
    static void jsections(std::ostringstream& o, const std::vector<Section>& v) {
        o << "[";
        for (size_t i = 0; i < v.size(); ++i) {
            if (i) o << ",";
            o << "{\"op\":"; jstr(o, v[i].op); o << ",\"dest\":"; jnode(o, v[i].dest); o << ",\"src\":[";
            for (size_t j = 0; j < v[i].src.size(); ++j) { if (j) o << ","; jnode(o, v[i].src[j]); }
            o << "]}";
        }
        o << "]";
    }
    static void jsegment(std::ostringstream& o, const Segment& s) {
        o << "{\"first\":"; jsections(o, s.first); o << ",\"i\":"; jsections(o, s.i); o << ",\"last\":"; jsections(o, s.last);
        o << ",\"tmp_used\":" << s.tmp_used << "}";
    }
    static void jints(std::ostringstream& o, const std::vector<i64>& v) { o << "["; for (size_t i = 0; i < v.size(); ++i) { if (i) o << ","; o << v[i]; } o << "]"; }
    static void jctx(std::ostringstream& o, const std::vector<PCCtx>& v) {
        o << "[";
        for (size_t i = 0; i < v.size(); ++i) {
            const PCCtx& c = v[i];
            if (i) o << ",";
            o << "{\"f_exp_id\":" << c.f_exp_id << ",\"t_exp_id\":" << c.t_exp_id << ",\"h1_id\":" << c.h1_id << ",\"h2_id\":" << c.h2_id << ",\"z_id\":" << c.z_id
              << ",\"c1_id\":" << c.c1_id << ",\"c2_id\":" << c.c2_id << ",\"num_id\":" << c.num_id << ",\"den_id\":" << c.den_id << "}";
        }
        o << "]";
    }
    template <class M> static void jmap(std::ostringstream& o, const M& m) {
        o << "{"; bool first = true;
        for (const char* s : SECTIONS) { if (!first) o << ","; first = false; jstr(o, s); o << ":" << m.at(s); }
        o << "}";
    }
    std::string to_json() const {
        std::ostringstream o;
        o << "{\"starkinfo\":{\"var_pol_map\":[";
        for (size_t i = 0; i < info.var_pol_map.size(); ++i) {
            const PolInfo& p = info.var_pol_map[i];
            if (i) o << ",";
            o << "{\"section\":"; jstr(o, p.section); o << ",\"section_pos\":" << p.section_pos << ",\"dim\":" << p.dim << ",\"exp_id\":" << p.exp_id << "}";
        }
        o << "],\"n_cm1\":" << info.n_cm1 << ",\"n_cm2\":" << info.n_cm2 << ",\"n_cm3\":" << info.n_cm3 << ",\"n_cm4\":" << info.n_cm4 << ",\"n_q\":" << info.n_q;
        o << ",\"pu_ctx\":"; jctx(o, info.pu_ctx); o << ",\"pe_ctx\":"; jctx(o, info.pe_ctx); o << ",\"ci_ctx\":"; jctx(o, info.ci_ctx);
        o << ",\"n_constants\":" << info.n_constants << ",\"n_publics\":" << info.n_publics << ",\"c_exp\":" << info.c_exp << ",\"im_exps\":{";
        { bool f = true; for (auto& kv : info.im_exps) { if (!f) o << ","; f = false; o << "\"" << kv.first << "\":" << (kv.second ? "true" : "false"); } }
        o << "},\"q_deg\":" << info.q_deg << ",\"q_dim\":" << info.q_dim << ",\"im_exps_list\":"; jints(o, info.im_exps_list);
        o << ",\"im_exp2cm\":{";
        { bool f = true; for (auto& kv : info.im_exp2cm) { if (!f) o << ","; f = false; o << "\"" << kv.first << "\":" << kv.second; } }
        o << "},\"qs\":"; jints(o, info.qs); o << ",\"exps_2ns\":[],\"exps_n\":[],\"ev_map\":[";
        for (size_t i = 0; i < info.ev_map.size(); ++i) { if (i) o << ","; jnode(o, info.ev_map[i]); }
        o << "],\"fri_exp_id\":" << info.fri_exp_id << ",\"n_exps\":" << info.n_exps;
        o << ",\"cm_n\":"; jints(o, info.cm_n); o << ",\"cm_2ns\":"; jints(o, info.cm_2ns); o << ",\"tmpexp_n\":"; jints(o, info.tmpexp_n);
        o << ",\"q_2ns\":"; jints(o, info.q_2ns); o << ",\"f_2ns\":"; jints(o, info.f_2ns);
        o << ",\"map_sections\":{";
        { bool f = true; for (const char* s : SECTIONS) { if (!f) o << ","; f = false; jstr(o, s); o << ":"; jints(o, info.map_sections.at(s)); } }
        o << "},\"map_sectionsN1\":"; jmap(o, info.map_sectionsN1); o << ",\"map_sectionsN3\":"; jmap(o, info.map_sectionsN3);
        o << ",\"map_sectionsN\":"; jmap(o, info.map_sectionsN); o << ",\"map_offsets\":"; jmap(o, info.map_offsets); o << ",\"map_deg\":"; jmap(o, info.map_deg);
        o << ",\"map_total_n\":" << info.map_total_n << ",\"exp2pol\":{";
        { bool f = true; for (auto& kv : info.exp2pol) { if (!f) o << ","; f = false; o << "\"" << kv.first << "\":" << kv.second; } }
        o << "},\"publics\":[";
        for (size_t i = 0; i < pil.publics.size(); ++i) {
            const Public& p = pil.publics[i];
            if (i) o << ",";
            o << "{\"polType\":"; jstr(o, p.polType); o << ",\"polId\":" << p.polId << ",\"idx\":" << p.idx << ",\"id\":" << p.id << ",\"name\":"; jstr(o, p.name); o << "}";
        }
        o << "],\"ev_idx\":{";
        auto evl = [&](const char* name, const std::vector<std::pair<std::pair<i64, i64>, i64>>& l) {
            jstr(o, name); o << ":[";
            for (size_t i = 0; i < l.size(); ++i) { if (i) o << ","; o << "[[" << l[i].first.first << "," << l[i].first.second << "]," << l[i].second << "]"; }
            o << "]";
        };
        evl("cm", info.ev_cm); o << ","; evl("const_", info.ev_const);
        o << "}},\"program\":{\"publics_code\":[";
        for (size_t i = 0; i < prog.publics_code.size(); ++i) { if (i) o << ","; jsegment(o, prog.publics_code[i]); }
        o << "],\"step2prev\":"; jsegment(o, prog.step2prev); o << ",\"step3prev\":"; jsegment(o, prog.step3prev); o << ",\"step3\":"; jsegment(o, prog.step3);
        o << ",\"step42ns\":"; jsegment(o, prog.step42ns); o << ",\"step52ns\":"; jsegment(o, prog.step52ns);
        o << ",\"verifier_code\":"; jsegment(o, prog.verifier_code); o << ",\"verifier_query_code\":"; jsegment(o, prog.verifier_query_code);
        o << "}}";
        return o.str();
    }
};

}  // namespace sgen

std::string starkinfo_generate(const std::string& pil_json, const std::string& stark_struct_json) {
    const JVal P = JParser::parse(pil_json.c_str()), S = JParser::parse(stark_struct_json.c_str());
    sgen::Gen g;
    g.load(P, S);
    g.run();
    return g.to_json();
}

}  // namespace zk

extern "C" char* zk_starkinfo_generate(const char* pil_json, const char* stark_struct_json) {
    try {
        if (!pil_json || !stark_struct_json) throw zk::Error("zk_starkinfo_generate: null argument");
        const std::string out = zk::starkinfo_generate(pil_json, stark_struct_json);
        char* p = (char*)malloc(out.size() + 1);
        if (!p) throw zk::Error("zk_starkinfo_generate: out of memory");
        memcpy(p, out.c_str(), out.size() + 1);
        return p;
    } catch (const std::exception& e) { zk::set_error(e.what()); return nullptr; }
    catch (...) { zk::set_error("unknown error"); return nullptr; }
}
