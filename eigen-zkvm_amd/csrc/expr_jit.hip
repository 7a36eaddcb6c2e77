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
#include <map>
#include <sstream>
#include <vector>

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
DEV f3 add31(f3 a, u64 b) { return f3{{gl::add(a.v[0], b), a.v[1], a.v[2]}}; }               // f3g.rs:338-341
DEV f3 add13(u64 a, f3 b) { return f3{{gl::add(b.v[0], a), b.v[1], b.v[2]}}; }               // f3g.rs:346-349
DEV f3 sub31(f3 a, u64 b) { return f3{{gl::sub(a.v[0], b), a.v[1], a.v[2]}}; }               // f3g.rs:381-384
DEV f3 sub13(u64 a, f3 b) { return f3{{gl::sub(a, b.v[0]), gl::neg(b.v[1]), gl::neg(b.v[2])}}; }  // f3g.rs:389-392
DEV f3 mul31(f3 a, u64 b) { return gl::f3_muls(a, b); }                                      // f3g.rs:412-416
DEV f3 mul13(u64 a, f3 b) { return gl::f3_muls(b, a); }                                      // f3g.rs:436-441
)ZKJIT";

struct Val { std::string name; int dim; };

struct Gen {
    std::ostringstream body;
    std::map<uint32_t, Val> tmp;                                   // tmp id -> current SSA value
    std::map<std::pair<uint32_t, uint32_t>, Val> fwd, fwd_prime;   // (buf, column) -> value this lane wrote at row i / i+next
    std::map<std::pair<uint32_t, uint32_t>, bool> written, prime_read;
    int n_val = 0;

    std::string fresh() { return "v" + std::to_string(n_val++); }

    Val load(const zk_operand& o) {
        ZK_REQUIRE(o.dim == 1 || o.dim == 3, "eval program: operand dim must be 1 or 3");
        std::ostringstream e;
        switch (o.kind) {
            case ZK_OPND_TMP: {
                auto it = tmp.find(o.id);
                ZK_REQUIRE(it != tmp.end(), "eval program: tmp read before write");
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
                    prime_read[key] = true;
                }
                e << "c.bufs[" << (int)o.buf << "] + " << (o.prime ? "ip" : "i") << " * " << o.stride << "ull + " << o.id;
                Val v{fresh(), o.dim};
                if (o.dim == 1) body << "    const u64 " << v.name << " = (" << e.str() << ")[0];\n";
                else            body << "    const f3 " << v.name << " = ld3(" << e.str() << ");\n";
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

    void store(const zk_operand& d, const Val& v) {
        if (d.kind == ZK_OPND_TMP) { tmp[d.id] = v; return; }                      // interpreter.rs:149-152
        ZK_REQUIRE(d.kind == ZK_OPND_MEM && d.buf < 16, "eval program: destination must be tmp or a section cell");
        // A primed destination (set_ref -> eval_map with prime, interpreter.rs:331-345) stores the value of row
        // i+next into row i+next's cell; the lane of that row stores the same field element there.
        std::ostringstream e;
        e << "(c.bufs[" << (int)d.buf << "] + " << (d.prime ? "ip" : "i") << " * " << d.stride << "ull + " << d.id << ")";
        if (v.dim == 1) body << "    " << e.str() << "[0] = " << v.name << ";\n";   // interpreter.rs:149-152
        else body << "    { u64* p = " << e.str() << "; p[0] = " << v.name << ".v[0]; p[1] = " << v.name << ".v[1]; p[2] = "
                  << v.name << ".v[2]; }\n";                                          // interpreter.rs:153-159
        auto key = std::make_pair((uint32_t)d.buf, d.id);
        if (d.prime) { fwd_prime[key] = v; return; }
        fwd[key] = v; written[key] = true;
    }

    void instr(const zk_instr& in) {
        if (in.op == ZK_OP_COPY) { store(in.dest, load(in.src[0])); return; }
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

}  // namespace
}  // namespace zk

using namespace zk;

struct zk_program {
    std::string source;
    std::vector<char> code;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    uint32_t n_instr = 0;
};

extern "C" {

zk_program_t* zk_program_compile(const zk_instr* code, uint32_t n_instr) {
    zk_program_t* p = nullptr;
    try {
        ZK_REQUIRE(code || n_instr == 0, "zk_program_compile: null code");
        p = new zk_program();
        p->n_instr = n_instr;
        Gen g;
        for (uint32_t k = 0; k < n_instr; ++k) g.instr(code[k]);
        for (auto& kv : g.prime_read)   // rows are evaluated concurrently: a column cannot be both written and read at i+next
            ZK_REQUIRE(!g.written.count(kv.first), "eval program: a column is written and read at the next row in the same step");
        std::ostringstream src;
        src << ZK_GL_JIT_SRC << JIT_HELPERS
            << "extern \"C\" __global__ __launch_bounds__(256) void zk_eval_kernel(const EvalCtx c, const u64 n, const u64 next) {\n"
            << "    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;\n"
            << "    if (i >= n) return;\n"
            << "    const u64 ip = (i + next) & (n - 1);\n"
            << g.body.str() << "}\n";
        p->source = src.str();

        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, p->source.c_str(), "zk_eval.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
            throw Error("hiprtcCreateProgram failed");
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
        hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
        if (rc != HIPRTC_SUCCESS) {
            std::string log = hiprtc_log(prog);
            hiprtcDestroyProgram(&prog);
            throw Error("hiprtc compile failed: " + log.substr(0, 2000));
        }
        size_t sz = 0; hiprtcGetCodeSize(prog, &sz);
        p->code.resize(sz);
        hiprtcGetCode(prog, p->code.data());
        hiprtcDestroyProgram(&prog);
        return p;
    } catch (const std::exception& e) { set_error(e.what()); delete p; return nullptr; }
}

const char* zk_program_source(const zk_program_t* p) { return p ? p->source.c_str() : ""; }

int zk_program_run_dev(zk_program_t* p, const zk_eval_ctx* ctx, uint32_t nbits_domain, uint64_t next, void* stream) {
    try {
        ZK_REQUIRE(p && ctx, "zk_program_run_dev: null");
        ZK_REQUIRE(nbits_domain <= 32, "zk_program_run_dev: domain too large");
        if (!p->module) {  // load lazily: compiling needs no GPU, running does
            ZK_HIP(hipModuleLoadData(&p->module, p->code.data()));
            ZK_HIP(hipModuleGetFunction(&p->fn, p->module, "zk_eval_kernel"));
        }
        struct { zk_eval_ctx c; uint64_t n; uint64_t next; } args{*ctx, 1ull << nbits_domain, next};
        size_t size = sizeof(args);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        const uint64_t blocks = (args.n + 255) / 256;
        ZK_HIP(hipModuleLaunchKernel(p->fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, cfg));
        return 0;
    } catch (const std::exception& e) { set_error(e.what()); return -1; }
}

int zk_program_free(zk_program_t* p) {
    if (p && p->module) (void)hipModuleUnload(p->module);
    delete p;
    return 0;
}

}  // extern "C"
