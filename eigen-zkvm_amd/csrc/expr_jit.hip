// Constraint-polynomial evaluation (the reference's "[XSLOW]" step, starky/README.md:34).
//
// Replaces starky/src/interpreter.rs:91-175 (Block::eval, a match-on-enum tree walker with string
// compares per operand), :187-225 (compile_code) and the per-chunk section copies of
// stark_gen.rs:786-963 (calculate_exps_parallel).
//
// Design: the prover program of one step (a list of three-address Sections -- the reference's
// Segment.first after fix_prover_code, starkinfo_codegen.rs:76-80) is translated ONCE into a
// straight-line HIP kernel and compiled for gfx950 with hipRTC; one lane evaluates one row.
// Temporaries become SSA values in VGPRs (the compiler does the register allocation), operand
// addresses `offset + ((i+next)%N)*size` (interpreter.rs:228-234) become immediate strides, and
// the F3G `dim` tag (f3g.rs:13-18) is resolved statically per value: add/sub/mul between dim-1 and
// dim-3 values pick the mixed forms of f3g.rs:323-449 at translation time.  Sections are read and
// written in place in HBM -- no per-thread context copies.
#include "zk_internal.h"
#include "../../include/zkgpu.h"
#include "gl_jit_src.h"
#include <hip/hiprtc.h>
#include "sha256.h"
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cerrno>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <map>
#include <set>
#include <sstream>
#include <tuple>
#include <vector>
extern "C" char** environ;                               // (for the compiler processes a setup starts: compile_spawned)

namespace zk {

namespace {

const char* JIT_HELPERS = R"ZKJIT(
using gl::f3;
#define DEV __device__ __forceinline__
struct EvalCtx {
    u64* bufs[16];
    const u64* publics; const u64* challenges; const u64* evals;
    const u64* x; const u64* zi; u64 zi_mask; const u64* xdiv; const u64* xdivw;
};
DEV f3 ld3(const u64* p) { return f3{{p[0], p[1], p[2]}}; }
// Section rows are staged through LDS one chunk of W columns at a time (W <= 19, LDS rows of W|1 words: odd, so the
// lanes' 8-byte reads of one column fall in distinct banks).  A wave moves the chunk of its 64 rows with consecutive lanes
// on consecutive words -- rows of 64 x W x 8 bytes in runs of W words -- instead of 64 lanes each walking its own row;
// row numbers wrap at the end of the domain (the primed rows of the last wave), rows past the launch's range are read
// all the same (they exist).  One wave, one buffer: LDS instructions of a wave execute in order, the fences only keep the
// compiler from moving them.  Stores stay per lane: a lane writes its row's cells back to back at the end of the kernel
// and L2 merges them into whole lines; staging them measured slower (DESIGN.md 3.3).
DEV void stage_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int W, int S>
DEV void stage_in(u64* __restrict__ wl, const u64* __restrict__ buf, unsigned rbase, unsigned mask, unsigned c0, unsigned lane) {
    constexpr unsigned WP = W | 1;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const unsigned w = lane + 64u * k, r = w / (unsigned)W, cc = w - r * (unsigned)W;
        wl[r * WP + cc] = buf[(u64)((rbase + r) & mask) * S + c0 + cc];
    }
    stage_sync();
}
DEV f3 add31(f3 a, u64 b) { return f3{{gl::add(a.v[0], b), a.v[1], a.v[2]}}; }               // f3g.rs:338-341
DEV f3 add13(u64 a, f3 b) { return f3{{gl::add(b.v[0], a), b.v[1], b.v[2]}}; }               // f3g.rs:346-349
DEV f3 sub31(f3 a, u64 b) { return f3{{gl::sub(a.v[0], b), a.v[1], a.v[2]}}; }               // f3g.rs:381-384
DEV f3 sub13(u64 a, f3 b) { return f3{{gl::sub(a, b.v[0]), gl::neg(b.v[1]), gl::neg(b.v[2])}}; }  // f3g.rs:389-392
DEV f3 mul31(f3 a, u64 b) { return gl::f3_muls(a, b); }                                      // f3g.rs:412-416
DEV f3 mul13(u64 a, f3 b) { return gl::f3_muls(b, a); }                                      // f3g.rs:436-441
// Horner chains over a challenge v with base-field terms, acc <- v * acc + d (the shape of the FRI and quotient
// polynomials' generated code), are evaluated as sum_j d_j * v^(e_j): field arithmetic is exact, so the value -- and its
// canonical word -- is the same, at 18 multiply-adds per term instead of a cubic-extension product.  pw holds the powers,
// each component split in three 22-bit limbs (two words); six accumulators per component take the 22x32-bit partial
// products without carries (up to 2^10 terms), one reduction per component at the end.
DEV void pacc(u64 (&A)[3][6], const u64* __restrict__ w, u64 d) {
    const unsigned x0 = (unsigned)d, x1 = (unsigned)(d >> 32);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const u64 w0 = w[2 * k], w1 = w[2 * k + 1];
        const unsigned l0 = (unsigned)w0, l1 = (unsigned)(w0 >> 32), l2 = (unsigned)w1;
        A[k][0] += (u64)l0 * x0; A[k][1] += (u64)l1 * x0; A[k][2] += (u64)l2 * x0;
        A[k][3] += (u64)l0 * x1; A[k][4] += (u64)l1 * x1; A[k][5] += (u64)l2 * x1;
    }
}
DEV u64 pfin1(const u64 (&a)[6]) {
    typedef unsigned __int128 u128;
    const u128 X0 = (u128)a[0] + ((u128)a[1] << 22) + ((u128)a[2] << 44);
    const u128 X1 = (u128)a[3] + ((u128)a[4] << 22) + ((u128)a[5] << 44);
    const u64 X1l = (u64)X1, X1h = (u64)(X1 >> 64);                       // X1 * 2^32 = X1l * 2^32 + X1h * 2^96 = X1l * 2^32 - X1h
    const u128 W = X0 + ((u128)X1l << 32) + (u128)(0xFFFFFFFF00000001ull - X1h);
    return gl::reduce_words((unsigned)W, (unsigned)(W >> 32), (unsigned)(W >> 64), (unsigned)(W >> 96));
}
DEV f3 pfin(const u64 (&A)[3][6]) { return f3{{pfin1(A[0]), pfin1(A[1]), pfin1(A[2])}}; }
// the power kernel (one lane per challenge, a few hundred extension products per launch) calls its products instead of inlining
// them: inlined, its one basic block of ~100 cubic-extension products took hipRTC 13 of the 16 s of step52ns's compilation
__device__ __noinline__ f3 f3_mul_call(f3 a, f3 b) { return gl::f3_mul(a, b); }
DEV void psplit(u64* __restrict__ o, f3 p) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const u64 x = p.v[k];
        o[2 * k] = (x & 0x3FFFFFull) | (((x >> 22) & 0x3FFFFFull) << 32);
        o[2 * k + 1] = x >> 44;
    }
}
)ZKJIT";

struct CTerm { std::string expr; int e; int sign; };       // a uniform cubic-extension operand (an eval) times v^e, added or subtracted
struct Val {
    std::string name; int dim;
    // ch != NOT_LAZY: not materialised; the value is sum_j terms[j].first * v^terms[j].second + sum_k +-cterms[k].expr * v^e
    // with v = challenge[ch]; ch == ANY_CH: no power of a challenge has been applied yet (all exponents are 0)
    static constexpr int NOT_LAZY = -1, ANY_CH = -2;
    int ch = NOT_LAZY;
    std::vector<std::pair<std::string, int>> terms;
    std::vector<CTerm> cterms;
    bool lazy() const { return ch != NOT_LAZY; }
};
struct ChainConst { int ch; std::vector<CTerm> cterms; };   // sum of a chain's uniform terms: computed once per launch

struct Gen {
    // Kernel text = prologue (every section read) + body (arithmetic) + epilogue (every section write).  A lane reads the
    // cells of its own rows only, and what it reads after writing comes from fwd, so all reads can be issued before the
    // first store and all stores after the last operation: nothing aliases a store any more, a row's cache lines are
    // fetched and written within a short window instead of once per use, and the reads of wide sections can go through
    // LDS (stage_in).  PoseidonG, 2^22 rows, kernels alone: 7.1 ms with reads and writes where the program has them,
    // 5.0 ms hoisted, 4.3 ms hoisted and staged.
    std::ostringstream body;
    // rows of at least this many words go through LDS (below that neighbouring lanes already share cache lines); 0 = never
    const uint32_t stage_min = getenv("ZK_JIT_STAGE_MIN") ? (uint32_t)atoi(getenv("ZK_JIT_STAGE_MIN")) : 8;
    // The prologue holds every word it reads in registers until its use: past HOIST_MAX words (PoseidonG's widest step
    // reads 93 and sits at the 256-register limit) the reads stay where the program has them and the compiler schedules them
    // -- still ahead of every store, which all sit in the epilogue.
    static constexpr uint32_t HOIST_MAX = 112;
    bool hoist_reads = true;
    struct Cell { uint32_t buf, id, dim, stride; bool prime; std::string name; };
    std::vector<Cell> reads, writes;    // in program order; name = the SSA value read into / written from
    static constexpr uint32_t CHUNK = 19;   // columns per staged chunk: 64 x 19 x 8 B per wave, four blocks of four waves per CU
    std::map<uint32_t, Val> tmp;                                   // tmp id -> current SSA value
    std::map<std::pair<uint32_t, uint32_t>, Val> fwd, fwd_prime;   // (buf, column) -> value this lane wrote at row i / i+next
    // Rows are evaluated concurrently (one lane per row), the reference evaluates them in order inside a chunk
    // (stark_gen.rs:752-783).  The two agree unless one row's lane reads a cell another row's lane writes: a write at row
    // i and a read of an overlapping cell range at row i+next, in either order.  Reads served from this lane's own
    // earlier store (fwd / fwd_prime) never reach memory and are not recorded.  Two writes of one cell (the code generator
    // stores an expression both at row i and, primed, at row i+next: t and t' of a plookup) carry the same field element
    // -- the same expression evaluated at the same row -- whichever lane lands last.
    struct Access { uint32_t buf, id, dim; bool prime; };
    std::vector<Access> mem_reads, mem_writes;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, bool>, Val> loaded;   // cells already read in the prologue
    static bool overlap(const Access& a, const Access& b) { return a.buf == b.buf && a.id < b.id + b.dim && b.id < a.id + a.dim; }
    void check_row_hazards() const {
        for (const Access& w : mem_writes) {
            for (const Access& r : mem_reads)
                ZK_REQUIRE(!(overlap(w, r) && w.prime != r.prime), "eval program: a column is written at one row and read at the next row in the same step");
        }
    }
    int n_val = 0;
    std::map<int, int> max_exp;                                     // challenge id -> highest power a materialised chain needs
    std::vector<ChainConst> chain_consts;

    std::string fresh() { return "v" + std::to_string(n_val++); }

    // emits the evaluation of a deferred Horner chain and turns v into an ordinary cubic-extension value
    void materialise(Val& v) {
        const std::string name = fresh();
        if (v.ch == Val::ANY_CH) {                              // never met a challenge: plain sums
            std::string d0 = v.terms[0].first;
            for (size_t j = 1; j < v.terms.size(); ++j) { const std::string t = fresh(); body << "    const u64 " << t << " = gl::add(" << d0 << ", " << v.terms[j].first << ");\n"; d0 = t; }
            std::string cur = fresh();
            body << "    const f3 " << cur << " = f3{{" << d0 << ", 0, 0}};\n";
            for (auto& ct : v.cterms) { const std::string t = fresh(); body << "    const f3 " << t << " = gl::f3_" << (ct.sign > 0 ? "add" : "sub") << "(" << cur << ", " << ct.expr << ");\n"; cur = t; }
            body << "    const f3 " << name << " = " << cur << ";\n";
        } else {
            // the carry-free accumulators of pacc hold 2^10 partial products of 54 bits: longer chains go in groups
            constexpr size_t GROUP = 512;
            int& mx = max_exp[v.ch];
            std::string sum;
            for (size_t g0 = 0; g0 < v.terms.size() || g0 == 0; g0 += GROUP) {
                const std::string acc = fresh();
                body << "    u64 " << acc << "[3][6] = {};\n";
                for (size_t j = g0; j < std::min(v.terms.size(), g0 + GROUP); ++j) {
                    mx = std::max(mx, v.terms[j].second);
                    body << "    pacc(" << acc << ", PW" << v.ch << "(" << v.terms[j].second << "), " << v.terms[j].first << ");\n";
                }
                const std::string part = "pfin(" + acc + ")";
                if (sum.empty()) sum = part;
                else { const std::string t = fresh(); body << "    const f3 " << t << " = gl::f3_add(" << sum << ", " << part << ");\n"; sum = t; }
            }
            if (v.cterms.empty()) body << "    const f3 " << name << " = " << sum << ";\n";
            else {
                for (auto& ct : v.cterms) mx = std::max(mx, ct.e);
                body << "    const f3 " << name << " = gl::f3_add(" << sum << ", ld3(KC(" << chain_consts.size() << ")));\n";
                chain_consts.push_back(ChainConst{v.ch, v.cterms});
            }
        }
        v.name = name; v.ch = Val::NOT_LAZY; v.terms.clear(); v.cterms.clear();
    }
    Val load(const zk_operand& o, bool keep_lazy = false) {
        ZK_REQUIRE(o.dim == 1 || o.dim == 3, "eval program: operand dim must be 1 or 3");
        switch (o.kind) {
            case ZK_OPND_TMP: {
                auto it = tmp.find(o.id);
                ZK_REQUIRE(it != tmp.end(), "eval program: tmp read before write");
                if (it->second.lazy() && !keep_lazy) materialise(it->second);   // the tmp keeps the materialised name for later readers
                return it->second;
            }
            case ZK_OPND_MEM: {
                ZK_REQUIRE(o.buf < 16, "eval program: buffer slot out of range");
                auto key = std::make_pair((uint32_t)o.buf, o.id);
                if (!o.prime) {
                    auto it = fwd.find(key);
                    if (it != fwd.end() && it->second.dim == o.dim) return it->second;
                } else {
                    auto it = fwd_prime.find(key);   // this lane computed the next-row value itself (e.g. t' of a plookup)
                    if (it != fwd_prime.end() && it->second.dim == o.dim) return it->second;
                }
                const Access acc{(uint32_t)o.buf, o.id, (uint32_t)o.dim, o.prime != 0};
                // the read is issued before this lane's stores: it must not need one of them (a store of another shape
                // over the same words; exact matches were served from fwd above)
                for (const Access& w : mem_writes)
                    ZK_REQUIRE(!(overlap(w, acc) && w.prime == acc.prime), "eval program: a read partially overlaps an earlier write of the same row");
                auto lkey = std::make_tuple(acc.buf, acc.id, acc.dim, acc.prime);
                auto lit = loaded.find(lkey);
                if (lit != loaded.end()) return lit->second;
                mem_reads.push_back(acc);
                Val v{fresh(), o.dim};
                if (!hoist_reads) {    // too many cells to hold from the top of the kernel: read at first use (still before every store)
                    const std::string at = "c.bufs[" + std::to_string(acc.buf) + "] + " + (acc.prime ? "ip" : "i") + " * " + std::to_string(o.stride) + "ull + " + std::to_string(acc.id);
                    if (o.dim == 1) body << "    const u64 " << v.name << " = (" << at << ")[0];\n";
                    else            body << "    const f3 " << v.name << " = ld3(" << at << ");\n";
                    loaded[lkey] = v;
                    return v;
                }
                reads.push_back(Cell{acc.buf, acc.id, acc.dim, o.stride, acc.prime, v.name});
                loaded[lkey] = v;
                return v;
            }
            case ZK_OPND_NUMBER: {
                ZK_REQUIRE(o.value < GL_P, "eval program: number not canonical");
                Val v{fresh(), 1};
                body << "    const u64 " << v.name << " = " << o.value << "ull;\n";
                return v;
            }
            case ZK_OPND_PUBLIC: { Val v{fresh(), 1}; body << "    const u64 " << v.name << " = c.publics[" << o.id << "];\n"; return v; }
            case ZK_OPND_CHALLENGE: { Val v{fresh(), 3}; body << "    const f3 " << v.name << " = ld3(c.challenges + " << 3 * o.id << ");\n"; return v; }
            case ZK_OPND_EVAL: { Val v{fresh(), 3}; body << "    const f3 " << v.name << " = ld3(c.evals + " << 3 * o.id << ");\n"; return v; }
            case ZK_OPND_X: { Val v{fresh(), 1}; body << "    const u64 " << v.name << " = c.x[i];\n"; return v; }
            case ZK_OPND_ZI: { Val v{fresh(), 1}; body << "    const u64 " << v.name << " = c.zi[i & c.zi_mask];\n"; return v; }
            case ZK_OPND_XDIVXSUBXI: { Val v{fresh(), 3}; body << "    const f3 " << v.name << " = ld3(c.xdiv + i * 3);\n"; return v; }
            case ZK_OPND_XDIVXSUBWXI: { Val v{fresh(), 3}; body << "    const f3 " << v.name << " = ld3(c.xdivw + i * 3);\n"; return v; }
            default: throw Error("eval program: unknown operand kind");
        }
    }

    void store(const zk_operand& d, Val v) {
        if (d.kind == ZK_OPND_TMP) { tmp[d.id] = v; return; }                      // interpreter.rs:149-152
        if (v.lazy()) materialise(v);
        ZK_REQUIRE(d.kind == ZK_OPND_MEM && d.buf < 16, "eval program: destination must be tmp or a section cell");
        // A primed destination (set_ref -> eval_map with prime, interpreter.rs:331-345) stores the value of row
        // i+next into row i+next's cell; the lane of that row stores the same field element there.
        writes.push_back(Cell{(uint32_t)d.buf, d.id, (uint32_t)v.dim, d.stride, d.prime != 0, v.name});   // interpreter.rs:149-159
        auto key = std::make_pair((uint32_t)d.buf, d.id);
        const Access acc{(uint32_t)d.buf, d.id, (uint32_t)v.dim, d.prime != 0};
        mem_writes.push_back(acc);
        auto& f = d.prime ? fwd_prime : fwd;
        for (auto it = f.begin(); it != f.end();) {     // an older value of other words of this store is no longer what the cells hold
            const Access old{it->first.first, it->first.second, (uint32_t)it->second.dim, acc.prime};
            if (it->first != key && overlap(old, acc)) it = f.erase(it); else ++it;
        }
        f[key] = v;
    }

    // one (section, row offset) group of cells, cut into chunks of at most CHUNK columns
    struct Group { uint32_t buf, stride; bool prime; std::vector<const Cell*> cells; };
    static std::vector<Group> groups_of(const std::vector<Cell>& cells) {
        std::vector<Group> g;
        for (const Cell& c : cells) {
            size_t k = 0;
            while (k < g.size() && !(g[k].buf == c.buf && g[k].prime == c.prime)) ++k;
            if (k == g.size()) g.push_back(Group{c.buf, c.stride, c.prime, {}});
            ZK_REQUIRE(g[k].stride == c.stride, "eval program: one section addressed with two row sizes");
            g[k].cells.push_back(&c);
        }
        return g;
    }
    static std::string word_of(const Cell& c, uint32_t j) { return c.dim == 1 ? c.name : c.name + ".v[" + std::to_string(j) + "]"; }
    uint32_t lds_words = 0;             // per wave
    std::string prologue() {
        std::ostringstream o;
        for (uint32_t dim : {1u, 3u}) {
            std::string names;
            for (const Cell& c : reads) if (c.dim == dim) names += (names.empty() ? "" : ", ") + c.name;
            if (!names.empty()) o << (dim == 1 ? "    u64 " : "    f3 ") << names << ";\n";
        }
        for (const Group& g : groups_of(reads)) {
            const uint32_t S = g.stride, n_chunks = (S + CHUNK - 1) / CHUNK, W0 = (S + n_chunks - 1) / n_chunks;
            const std::string row = g.prime ? "ip" : "i", rbase = g.prime ? "i0 + (unsigned)next" : "i0";
            for (uint32_t c0 = 0; c0 < S; c0 += W0) {
                const uint32_t W = std::min(W0, S - c0);
                uint32_t used = 0;
                for (uint32_t col = c0; col < c0 + W; ++col) {
                    bool u = false;
                    for (const Cell* c : g.cells) u = u || (c->id <= col && col < c->id + c->dim);
                    used += u;
                }
                if (!used) continue;
                const bool staged = stage_min && S >= stage_min && 2 * used >= W;
                if (staged) {
                    lds_words = std::max(lds_words, 64 * (W | 1));
                    o << "    stage_in<" << W << ", " << S << ">(wl, c.bufs[" << g.buf << "], " << rbase << ", (unsigned)(n - 1), " << c0 << ", lane);\n";
                }
                for (const Cell* c : g.cells)
                    for (uint32_t j = 0; j < c->dim; ++j) {
                        const uint32_t col = c->id + j;
                        if (col < c0 || col >= c0 + W) continue;
                        if (staged) o << "    " << word_of(*c, j) << " = wl[lane * " << (W | 1) << " + " << col - c0 << "];\n";
                        else o << "    " << word_of(*c, j) << " = c.bufs[" << g.buf << "][" << row << " * " << S << "ull + " << col << "];\n";
                    }
                if (staged) o << "    stage_sync();\n";
            }
        }
        return o.str();
    }
    std::string epilogue() {
        std::ostringstream o;
        o << "    if (live) {\n";
        for (const Cell& c : writes)          // program order: a later store of the same word wins
            for (uint32_t j = 0; j < c.dim; ++j)
                o << "        c.bufs[" << c.buf << "][" << (c.prime ? "ip" : "i") << " * " << c.stride << "ull + " << c.id + j << "] = " << word_of(c, j) << ";\n";
        o << "    }\n";
        return o.str();
    }

    void instr(const zk_instr& in) {
        if (in.op == ZK_OP_COPY) { store(in.dest, load(in.src[0], in.dest.kind == ZK_OPND_TMP)); return; }
        // Horner steps over a challenge stay symbolic: v * (base-field value | chain), chain + base-field value, chain + chain,
        // and the FRI polynomial's terms (column - eval): a base-field value minus a uniform cubic-extension operand
        if (in.op == ZK_OP_MUL || in.op == ZK_OP_ADD || in.op == ZK_OP_SUB) {
            const bool c0 = in.src[0].kind == ZK_OPND_CHALLENGE, c1 = in.src[1].kind == ZK_OPND_CHALLENGE;
            auto compatible = [](const Val& x, const Val& y) { return x.ch == y.ch || x.ch == Val::ANY_CH || y.ch == Val::ANY_CH; };
            auto merge = [](Val x, const Val& y) {
                if (x.ch == Val::ANY_CH) x.ch = y.ch;
                x.terms.insert(x.terms.end(), y.terms.begin(), y.terms.end());
                x.cterms.insert(x.cterms.end(), y.cterms.begin(), y.cterms.end());
                return x;
            };
            if (in.op == ZK_OP_MUL && c0 != c1) {
                const int ch = (int)(c0 ? in.src[0] : in.src[1]).id;
                Val o = load(c0 ? in.src[1] : in.src[0], true);
                if (o.lazy() && (o.ch == ch || o.ch == Val::ANY_CH)) {
                    o.ch = ch;
                    for (auto& t : o.terms) ++t.second;
                    for (auto& t : o.cterms) ++t.e;
                    store(in.dest, o); return;
                }
                if (!o.lazy() && o.dim == 1) { Val r{"", 3}; r.ch = ch; r.terms.push_back({o.name, 1}); store(in.dest, r); return; }
            } else if (in.op == ZK_OP_ADD && !c0 && !c1) {
                Val a = load(in.src[0], true), b = load(in.src[1], true);
                if (a.lazy() && !b.lazy() && b.dim == 1) { a.terms.push_back({b.name, 0}); store(in.dest, a); return; }
                if (b.lazy() && !a.lazy() && a.dim == 1) { b.terms.push_back({a.name, 0}); store(in.dest, b); return; }
                if (a.lazy() && b.lazy() && compatible(a, b)) { store(in.dest, merge(a, b)); return; }
            } else if (in.op == ZK_OP_SUB && in.src[1].kind == ZK_OPND_EVAL && in.src[0].kind != ZK_OPND_CHALLENGE) {
                Val a = load(in.src[0], true);
                if (!a.lazy() && a.dim == 1) {
                    Val r{"", 3}; r.ch = Val::ANY_CH; r.terms.push_back({a.name, 0});
                    r.cterms.push_back(CTerm{"ld3(c.evals + " + std::to_string(3 * in.src[1].id) + ")", 0, -1});
                    store(in.dest, r); return;
                }
            }
        }
        Val a = load(in.src[0]), b = load(in.src[1]);
        const char* fn = in.op == ZK_OP_ADD ? "add" : in.op == ZK_OP_SUB ? "sub" : in.op == ZK_OP_MUL ? "mul" : nullptr;
        ZK_REQUIRE(fn, "eval program: unknown op");
        Val r{fresh(), (a.dim == 3 || b.dim == 3) ? 3 : 1};
        if (r.dim == 1) body << "    const u64 " << r.name << " = gl::" << fn << "(" << a.name << ", " << b.name << ");\n";
        else if (a.dim == 3 && b.dim == 3) body << "    const f3 " << r.name << " = gl::f3_" << fn << "(" << a.name << ", " << b.name << ");\n";
        else body << "    const f3 " << r.name << " = " << fn << (a.dim == 3 ? "31" : "13") << "(" << a.name << ", " << b.name << ");\n";
        store(in.dest, r);
    }
};

std::string hiprtc_log(hiprtcProgram prog) {
    size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    return log;
}


// ---- code objects of the generated kernels: compiled once per text ---------------------------------------------------------
// hipRTC takes seconds per step program (PoseidonG's five: most of a setup), and the text is a pure function of the PIL, the
// StarkStruct and this file.  Code objects are therefore kept (a) in the process, keyed by sha256(compiler + runtime version |
// options | text) -- the worker setups of one process compile each program once, and different programs compile side by side
// (a setup starts its step programs on one host thread each) -- and (b) on disk under $ZK_JIT_CACHE (default
// $XDG_CACHE_HOME/zkgpu or ~/.cache/zkgpu; "off" disables), one file per key: a 48-byte header (magic, payload length, sha256 of
// the payload) + the code object, written to a temporary name and renamed.  A file whose header or digest does not match is
// ignored and replaced (the digest sits in the same file: it catches truncation and bit rot, NOT a planted file -- whoever can write
// the file can write a matching digest).  What keeps foreign code objects out is ownership: the cache directory AND every directory
// above it up to $HOME (or /) must be the user's own or root's and not writable by group or others (sticky /tmp-style directories
// excepted), and a cache file is opened with O_NOFOLLOW and must be a regular file owned by this user with no group / other write
// bit; anything else is ignored.  An object from disk that the driver refuses to load is deleted and compiled again
// (zk_program_run_rows_dev).
int jit_waves() {                                      // minimum waves per SIMD asked of the step kernels (launch bounds)
    static const int w = [] { const char* e = getenv("ZK_JIT_WAVES"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v > 8 ? 8 : v; }();
    return w;
}
std::mutex g_jit_mu;
std::condition_variable g_jit_cv;
struct CodeObj { std::vector<char> bytes; std::string key; bool from_disk = false; };
std::map<std::string, std::shared_ptr<const CodeObj>> g_jit_mem;
std::set<std::string> g_jit_inflight;                     // keys some thread is reading or compiling right now
JitStats g_jit_stats;
const char JIT_MAGIC[8] = {'Z', 'K', 'C', 'O', '0', '0', '0', '1'};

std::string jit_cache_dir() {
    const char* e = getenv("ZK_JIT_CACHE");
    if (e && (!strcmp(e, "off") || !strcmp(e, "0") || !*e)) return "";
    if (e) return e;
    if (const char* x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/zkgpu";
    if (const char* h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/zkgpu";
    return "";
}
void mkdirs(const std::string& d) {
    for (size_t i = 1; i <= d.size(); ++i)
        if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), 0700);
}
// code objects are executed: only a directory of the user's own that nobody else can write to is trusted
bool jit_dir_trusted(const std::string& dir) {
    auto warn = [&](const std::string& which, const char* why) {
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "[zkgpu] code-object cache %s: %s %s: not used\n", dir.c_str(), which.c_str(), why); }
        return false;
    };
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;               // (a symlink in place of the directory is not followed)
    if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return warn(dir, "is not owned by this user or is writable by others");
    // the directories above: whoever can rename or replace one of them can swap the cache.  Walk up to $HOME (the user's own, checked like the
    // leaf) or to /; a parent must belong to this user or to root, and must not be writable by group / others unless it is sticky (/tmp)
    const char* home = getenv("HOME");
    std::string p = dir;
    while (p.size() > 1) {
        const size_t k = p.rfind('/');
        p = k == 0 || k == std::string::npos ? "/" : p.substr(0, k);
        if (stat(p.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
        const bool sticky = st.st_mode & S_ISVTX;
        if ((st.st_uid != geteuid() && st.st_uid != 0) || ((st.st_mode & (S_IWGRP | S_IWOTH)) && !sticky)) return warn(p, "(above the cache) belongs to another user or is writable by others");
        if (home && p == home) break;
    }
    return true;
}
std::string jit_path(const std::string& key) {
    const std::string dir = jit_cache_dir();
    return dir.empty() ? "" : dir + "/" + key + ".co";
}
bool jit_read(const std::string& path, std::vector<char>& out) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);                 // the file itself: ours, regular, nobody else may write it
    if (fd < 0) return false;
    struct stat fst;
    if (fstat(fd, &fst) != 0 || !S_ISREG(fst.st_mode) || fst.st_uid != geteuid() || (fst.st_mode & (S_IWGRP | S_IWOTH))) { close(fd); return false; }
    FILE* f = fdopen(fd, "rb");
    if (!f) { close(fd); return false; }
    std::vector<char> buf;
    char tmp[65536]; size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    if (buf.size() < 48 + 16 || memcmp(buf.data(), JIT_MAGIC, 8) != 0) return false;
    uint64_t len = 0; memcpy(&len, buf.data() + 8, 8);
    if (len != buf.size() - 48 || memcmp(buf.data() + 48, "\x7f" "ELF", 4) != 0) return false;   // truncated or foreign
    const std::string h = sha256_hex(buf.data() + 48, (size_t)len);
    char hex[65] = {0};
    for (int i = 0; i < 32; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)buf[16 + i]);
    if (h != hex) return false;                                                                 // truncation or bit rot (not a defence against a planted file: see above)
    out.assign(buf.begin() + 48, buf.end());
    return true;
}
void jit_write(const std::string& dir, const std::string& path, const std::vector<char>& code) {
    mkdirs(dir);                                          // best effort: a read-only or missing cache directory costs nothing but the next compile
    if (!jit_dir_trusted(dir)) return;
    const std::string h = sha256_hex(code.data(), code.size());
    char head[48]; memcpy(head, JIT_MAGIC, 8);
    const uint64_t len = code.size(); memcpy(head + 8, &len, 8);
    for (int i = 0; i < 32; ++i) { unsigned b = 0; sscanf(h.c_str() + 2 * i, "%2x", &b); head[16 + i] = (char)b; }
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string((unsigned long)(uintptr_t)&head);
    if (FILE* f = fopen(tmp.c_str(), "wb")) {
        const bool ok = fwrite(head, 1, 48, f) == 48 && fwrite(code.data(), 1, code.size(), f) == code.size();
        if (fclose(f) == 0 && ok) { if (rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str()); }
        else (void)remove(tmp.c_str());
    }
}

// hipRTC (comgr) compiles one program at a time per process, whatever the number of host threads (measured: three step programs on
// three threads take what they take one after the other, profiles/r04/cold_setup.txt).  A setup that compiles several programs at once
// therefore hands each to a helper process, `zkgpu_jitc` next to libzkgpu.so (csrc/jitc_main.cpp: the same hipRTC call, the same
// options, the same code object): set per thread by jit_prefer_spawn().  0 = compiled, > 0 = the compiler rejected the text (log),
// < 0 = no helper or it could not be started (the caller compiles in process).
thread_local bool t_jit_spawn = false;
std::string jitc_path() {
    if (const char* e = getenv("ZK_JITC")) return (!strcmp(e, "off") || !*e) ? "" : e;
    Dl_info info;
    if (!dladdr((const void*)&jitc_path, &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    p = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/zkgpu_jitc";
    return access(p.c_str(), X_OK) == 0 ? p : "";
}
int compile_spawned(const std::string& source, const char* const* opts, int n_opts, std::vector<char>& out, std::string& log) {
    static const std::string helper = jitc_path();
    if (helper.empty()) return -1;
    const char* td = getenv("TMPDIR");
    std::string dir = std::string(td && *td ? td : "/tmp") + "/zkgpu_jitc_XXXXXX";
    if (!mkdtemp(&dir[0])) return -1;                     // 0700, ours
    const std::string src = dir + "/k.hip", obj = dir + "/k.co", err = dir + "/k.log";
    struct Cleanup { const std::string &a, &b, &c, &d; ~Cleanup() { (void)remove(a.c_str()); (void)remove(b.c_str()); (void)remove(c.c_str()); (void)rmdir(d.c_str()); } } cleanup{src, obj, err, dir};
    {
        FILE* f = fopen(src.c_str(), "wb");
        if (!f) return -1;
        const bool ok = fwrite(source.data(), 1, source.size(), f) == source.size();
        if (fclose(f) != 0 || !ok) return -1;
    }
    std::vector<char*> argv{const_cast<char*>(helper.c_str()), const_cast<char*>(src.c_str()), const_cast<char*>(obj.c_str())};
    for (int i = 0; i < n_opts; ++i) argv.push_back(const_cast<char*>(opts[i]));
    argv.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    if (posix_spawn_file_actions_init(&fa) != 0) return -1;
    (void)posix_spawn_file_actions_addopen(&fa, 2, err.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    (void)posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return -1;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0) if (errno != EINTR) return -1;
    auto slurp = [](const std::string& p, std::vector<char>& v) {
        FILE* f = fopen(p.c_str(), "rb"); if (!f) return false;
        char tmp[65536]; size_t n; v.clear();
        while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) v.insert(v.end(), tmp, tmp + n);
        fclose(f); return true;
    };
    if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && slurp(obj, out) && out.size() > 16 && !memcmp(out.data(), "\x7f" "ELF", 4)) return 0;
    std::vector<char> l;
    if (WIFEXITED(status) && WEXITSTATUS(status) == 1 && slurp(err, l) && !l.empty()) { log.assign(l.begin(), l.end()); return 1; }   // the compiler's verdict on the text (jitc_main.cpp: 1 = hiprtcCompileProgram rejected)
    return -1;                                            // the helper itself failed (status 3: I/O, a full $TMPDIR, hiprtcCreateProgram; a missing library; a signal): compile in process
}

std::shared_ptr<const CodeObj> compile_cached(const std::string& source, bool skip_disk = false) {
    static const char* const opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    int maj = 0, min = 0, rt = 0; (void)hiprtcVersion(&maj, &min);
    if (hipRuntimeGetVersion(&rt) != hipSuccess) { (void)hipGetLastError(); rt = 0; }
    // the compiler behind hipRTC changes with patch releases too: the runtime's full version number and the headers' are part of the key
    std::string keyed = "hiprtc " + std::to_string(maj) + "." + std::to_string(min) + " rt " + std::to_string(rt) + " hip " + std::to_string(HIP_VERSION);
    for (const char* o : opts) { keyed += ' '; keyed += o; }
    keyed += '\n'; keyed += source;
    const std::string key = sha256_hex(keyed.data(), keyed.size());
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    {   // the same text once at a time (a second setup of the same circuit waits and then hits); different texts side by side
        std::unique_lock<std::mutex> lk(g_jit_mu);
        for (;;) {
            auto it = g_jit_mem.find(key);
            if (it != g_jit_mem.end()) { g_jit_stats.mem_hits++; return it->second; }
            if (!g_jit_inflight.count(key)) break;
            g_jit_cv.wait(lk);
        }
        g_jit_inflight.insert(key);
    }
    struct Done { const std::string& k; ~Done() { std::lock_guard<std::mutex> lk(g_jit_mu); g_jit_inflight.erase(k); g_jit_cv.notify_all(); } } done{key};
    auto sp = std::make_shared<CodeObj>();
    sp->key = key;
    const std::string dir = jit_cache_dir(), path = jit_path(key);
    if (!path.empty() && !skip_disk && jit_dir_trusted(dir) && jit_read(path, sp->bytes)) {
        sp->from_disk = true;
        std::lock_guard<std::mutex> lk(g_jit_mu);
        g_jit_mem[key] = sp; g_jit_stats.disk_hits++; g_jit_stats.ms += ms_since();
        return sp;
    }
    bool spawned = false;
    static const bool always_spawn = getenv("ZK_JIT_SPAWN") != nullptr;   // (tests: every compilation through the helper)
    if (t_jit_spawn || always_spawn) {                    // a setup compiling its programs side by side: one hipRTC process each
        std::string log;
        const int r = compile_spawned(source, opts, 3, sp->bytes, log);
        if (r > 0) throw Error("hiprtc compile failed: " + log.substr(0, 2000));
        spawned = r == 0;                                 // r < 0: no helper / could not start it -> in process
        if (spawned) { std::lock_guard<std::mutex> lk(g_jit_mu); g_jit_stats.spawned++; }
    }
    if (!spawned) {
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, source.c_str(), "zk_eval.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
            throw Error("hiprtcCreateProgram failed");
        hiprtcResult rc = hiprtcCompileProgram(prog, 3, const_cast<const char**>(opts));
        if (rc != HIPRTC_SUCCESS) {
            std::string log = hiprtc_log(prog);
            hiprtcDestroyProgram(&prog);
            throw Error("hiprtc compile failed: " + log.substr(0, 2000));
        }
        size_t sz = 0; hiprtcGetCodeSize(prog, &sz);
        sp->bytes.resize(sz);
        hiprtcGetCode(prog, sp->bytes.data());
        hiprtcDestroyProgram(&prog);
    }
    if (!path.empty()) jit_write(dir, path, sp->bytes);
    std::lock_guard<std::mutex> lk(g_jit_mu);
    g_jit_mem[key] = sp; g_jit_stats.compiled++; g_jit_stats.ms += ms_since();
    return sp;
}
// an object that came from disk and does not load: the file and the in-memory entry go, the text is compiled again
std::shared_ptr<const CodeObj> recompile_after_bad_load(const std::shared_ptr<const CodeObj>& bad, const std::string& source) {
    {
        std::lock_guard<std::mutex> lk(g_jit_mu);
        auto it = g_jit_mem.find(bad->key);
        if (it != g_jit_mem.end() && it->second == bad) g_jit_mem.erase(it);
    }
    const std::string path = jit_path(bad->key);
    if (!path.empty()) (void)remove(path.c_str());
    return compile_cached(source, true);
}
}  // namespace
void jit_prefer_spawn(bool on) { t_jit_spawn = on; }
JitStats jit_stats() { std::lock_guard<std::mutex> lk(g_jit_mu); return g_jit_stats; }
}  // namespace zk

using namespace zk;

struct zk_program {
    std::string source;
    std::shared_ptr<const CodeObj> code;   // the code object, shared by every program of this process with the same text
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr, fn_pow = nullptr;
    uint32_t n_instr = 0;
    uint32_t pow_entries = 0;      // (challenge, exponent) pairs of the power table, 6 words each
    void* d_pow = nullptr;
};

extern "C" {

zk_program_t* zk_program_compile(const zk_instr* code, uint32_t n_instr) {
    zk_program_t* p = nullptr;
    bind_device();
    try {
        ZK_REQUIRE(code || n_instr == 0, "zk_program_compile: null code");
        p = new zk_program();
        p->n_instr = n_instr;
        Gen g;
        {   // words the program reads from sections (an upper bound: reads of cells it wrote itself never reach memory)
            std::set<std::tuple<uint32_t, uint32_t, bool>> words;
            for (uint32_t k = 0; k < n_instr; ++k)
                for (const zk_operand& o : code[k].src)
                    if (o.kind == ZK_OPND_MEM) for (uint32_t j = 0; j < o.dim && j < 3; ++j) words.insert(std::make_tuple((uint32_t)o.buf, o.id + j, o.prime != 0));
            g.hoist_reads = words.size() <= Gen::HOIST_MAX;
        }
        for (uint32_t k = 0; k < n_instr; ++k) g.instr(code[k]);
        g.check_row_hazards();
        std::ostringstream src, powk;
        src << ZK_GL_JIT_SRC << JIT_HELPERS;
        // the power table: one lane per challenge writes v^0 .. v^max, split for pacc, then the constants of the chains on v
        uint32_t off = 0, lane = 0;
        std::map<int, uint32_t> off_of;
        for (auto& kv : g.max_exp) { off_of[kv.first] = off; off += (uint32_t)kv.second + 1; }
        const uint32_t pow_words = off * 6;
        src << "#define KC(n) (pw + " << pow_words << " + 3 * (n))\n";
        for (auto& kv : g.max_exp) {
            const uint32_t o = off_of[kv.first];
            src << "#define PW" << kv.first << "(e) (pw + (" << o << " + (e)) * 6)\n";
            powk << "    if (threadIdx.x == " << lane++ << ") {\n        const f3 v = ld3(c.challenges + " << 3 * kv.first << "); f3 P[" << kv.second + 2 << "]; P[0] = f3{{1, 0, 0}};\n"
                 << "#pragma unroll 1\n        for (int e = 0; e <= " << kv.second << "; ++e) { psplit(pw + (" << o << " + e) * 6, P[e]); P[e + 1] = f3_mul_call(P[e], v); }\n";
            for (size_t n = 0; n < g.chain_consts.size(); ++n) {
                if (g.chain_consts[n].ch != kv.first) continue;
                powk << "        { f3 k = f3{{0, 0, 0}};\n";
                for (auto& ct : g.chain_consts[n].cterms)
                    powk << "          k = gl::f3_" << (ct.sign > 0 ? "add" : "sub") << "(k, f3_mul_call(" << ct.expr << ", P[" << ct.e << "]));\n";
                powk << "          u64* o = pw + " << pow_words << " + 3 * " << n << "; o[0] = k.v[0]; o[1] = k.v[1]; o[2] = k.v[2]; }\n";
            }
            powk << "    }\n";
        }
        ZK_REQUIRE(lane <= 64, "eval program: too many challenges with Horner chains");
        std::string pro_src = g.prologue(), epi_src = g.epilogue();
        if (g.lds_words) pro_src = "    __shared__ u64 zk_stage[4][" + std::to_string(g.lds_words) + "];\n    u64* const wl = zk_stage[wave];\n" + pro_src;
        p->pow_entries = (pow_words + 3 * (uint32_t)g.chain_consts.size() + 5) / 6;
        src << "extern \"C\" __global__ __launch_bounds__(64) void zk_pow_kernel(const EvalCtx c, u64* __restrict__ pw) {\n" << powk.str() << "}\n"
            // (256, W): at least W waves per SIMD.  A step program of a wide PIL keeps a row's every operand live -- 280 VGPRs for the compressor-shaped
            // circuit's step42ns: ONE wave per SIMD, and a lone wave issues a dependent instruction every 8.3 cycles where two waves fill the 4.2-cycle issue
            // slots (profiles/r05/ubench_lat.txt).  W = 2 caps the kernel at 256 registers (24 spills in that kernel); ZK_JIT_WAVES overrides (1 = no cap).
            << "extern \"C\" __global__ __launch_bounds__(256" << (jit_waves() > 1 ? ", " + std::to_string(jit_waves()) : std::string()) << ") void zk_eval_kernel(const EvalCtx c, const u64 n, const u64 next, const u64* __restrict__ pw, const u64 row0, const u64 count) {\n"
            << "    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;\n"
            << "    const u64 k0 = (u64)blockIdx.x * blockDim.x + wave * 64u;\n"
            << "    if (k0 >= count) return;                                  // whole wave past the range\n"
            << "    const unsigned i0 = (unsigned)(row0 + k0);                    // first row of this wave (the domain has at most 2^32 rows)\n"
            << "    const bool live = k0 + lane < count;                        // lanes past the range compute on rows that exist and store nothing\n"
            << "    const u64 i = (u64)((i0 + lane) & (unsigned)(n - 1));\n"
            << "    const u64 ip = (i + next) & (n - 1);\n"
            << pro_src << g.body.str() << epi_src << "}\n";
        p->source = src.str();

        p->code = compile_cached(p->source);
        return p;
    } catch (const std::exception& e) { set_error(e.what()); delete p; return nullptr; }
}

const char* zk_program_source(const zk_program_t* p) { return p ? p->source.c_str() : ""; }

void zk_jit_cache_stats(uint64_t out[3]) {
    const JitStats j = jit_stats();
    out[0] = j.compiled; out[1] = j.disk_hits; out[2] = j.mem_hits;
}

int zk_program_run_dev(zk_program_t* p, const zk_eval_ctx* ctx, uint32_t nbits_domain, uint64_t next, void* stream) {
    return zk_program_run_rows_dev(p, ctx, nbits_domain, next, 0, nbits_domain <= 32 ? 1ull << nbits_domain : 0, stream);
}

int zk_program_run_rows_dev(zk_program_t* p, const zk_eval_ctx* ctx, uint32_t nbits_domain, uint64_t next, uint64_t row0, uint64_t count,
                            void* stream) {
    bind_device();
    try {
        ZK_REQUIRE(p && ctx, "zk_program_run_dev: null");
        ZK_REQUIRE(nbits_domain <= 32, "zk_program_run_dev: domain too large");
        ZK_REQUIRE(row0 <= (1ull << nbits_domain) && count <= (1ull << nbits_domain) - row0, "zk_program_run_rows_dev: rows outside the domain");
        if (count == 0) return 0;
        if (!p->module) {  // load lazily: compiling needs no GPU, running does
            auto load = [&]() -> hipError_t {
                hipError_t e = hipModuleLoadData(&p->module, p->code->bytes.data());
                if (e == hipSuccess) e = hipModuleGetFunction(&p->fn, p->module, "zk_eval_kernel");
                if (e == hipSuccess) e = hipModuleGetFunction(&p->fn_pow, p->module, "zk_pow_kernel");
                if (e != hipSuccess) { (void)hipGetLastError(); if (p->module) { (void)hipModuleUnload(p->module); p->module = nullptr; } p->fn = p->fn_pow = nullptr; }
                return e;
            };
            hipError_t e = load();
            if (e != hipSuccess && p->code->from_disk) {   // a cached file the driver refuses: drop it, compile the text again, once
                p->code = recompile_after_bad_load(p->code, p->source);
                e = load();
            }
            if (e != hipSuccess) throw Error(std::string("loading a constraint kernel's code object failed: ") + hipGetErrorString(e));
            if (p->pow_entries) ZK_HIP(hipMalloc(&p->d_pow, (size_t)p->pow_entries * 48));
        }
        if (p->pow_entries) {   // powers of this run's challenges, on the same stream
            struct { zk_eval_ctx c; void* pw; } pa{*ctx, p->d_pow};
            size_t psz = sizeof(pa);
            void* pcfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psz, HIP_LAUNCH_PARAM_END};
            ZK_HIP(hipModuleLaunchKernel(p->fn_pow, 1, 1, 1, 64, 1, 1, 0, on_stream((hipStream_t)stream), nullptr, pcfg));
        }
        struct { zk_eval_ctx c; uint64_t n; uint64_t next; const void* pw; uint64_t row0; uint64_t count; } args{*ctx, 1ull << nbits_domain, next, p->d_pow, row0, count};
        size_t size = sizeof(args);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        const uint64_t blocks = (count + 255) / 256;
        ZK_HIP(hipModuleLaunchKernel(p->fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, on_stream((hipStream_t)stream), nullptr, cfg));
        return 0;
    } catch (const std::exception& e) { set_error(e.what()); return -1; }
}

int zk_program_free(zk_program_t* p) {
    bind_device();
    if (p && p->d_pow) (void)hipFree(p->d_pow);
    if (p && p->module) (void)hipModuleUnload(p->module);
    delete p;
    return 0;
}

}  // extern "C"
