// compressor12 exec on gfx950 -- SURVEY.md 8(f)-4: recursion/src/compressor12/compressor12_exec.rs:58-103, the step of
// `recursive_proof_to_snark.sh` between the circom witness and the committed trace of the recursive STARK:
//   w <- witness;  for every PlonkAdd (a, b, ca, cb): w.push(w[a] * ca + w[b] * cb)          (:60-66, sequential)
//   cm[i][c] = s_map[i][c] != 0 ? w[s_map[i][c]] : 0  for i < s_map_column_len, 0 below, c < 12  (:72-94)
// so that the .cm matrix is born in HBM, where zk_stark_gen_dev takes it.
// The .exec file (compressor12_setup.rs:51-83 / compressor12_exec.rs:110-125) is a JSON array of u64:
// [adds_len, s_map_column_len, adds (4 words each), s_map (row i, column c at 12 i + c)]; the two coefficients of an add
// are the RAW words of an FGL (field_gl.rs:503-507, read back with from_raw_repr, :337-347), i.e. value * 2^64 mod p.
// The additions form a DAG (an add may read earlier sums): they are grouped once by depth, one launch per depth.
#include "zk_internal.h"
#include <algorithm>
#include <memory>
#include <cstring>
#include <vector>

namespace zk {
namespace {

constexpr u64 GLP = 0xFFFFFFFF00000001ULL;
u64 host_mulmod(u64 a, u64 b) { return (u64)((unsigned __int128)a * b % GLP); }
u64 host_powmod(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = host_mulmod(r, a); a = host_mulmod(a, a); e >>= 1; } return r; }

// the whole file is digits and punctuation: a JVal per number would cost gigabytes for a real circuit
std::vector<u64> parse_u64_array(const char* s, size_t n) {
    std::vector<u64> out;
    size_t i = 0;
    while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i;
    if (i >= n || s[i] != '[') throw std::runtime_error("exec file: a JSON array of integers is expected");
    ++i;
    bool closed = false;
    while (i < n) {
        const char c = s[i];
        if (c >= '0' && c <= '9') {
            u64 v = 0;
            while (i < n && s[i] >= '0' && s[i] <= '9') {
                const u64 d = (u64)(s[i] - '0');
                if (v > (~0ull - d) / 10) throw std::runtime_error("exec file: integer does not fit 64 bits");
                v = v * 10 + d; ++i;
            }
            out.push_back(v);
        } else if (c == ',' || c == ' ' || c == '\n' || c == '\t' || c == '\r') ++i;
        else if (c == ']') { closed = true; break; }
        else throw std::runtime_error("exec file: unexpected character in the array");
    }
    if (!closed) throw std::runtime_error("exec file: unterminated array");
    return out;
}

struct AddOp { u32 a, b, dst, pad; u64 ca, cb; };   // canonical coefficients

__global__ __launch_bounds__(256) void c12_load_kernel(const u64* __restrict__ wit, u64 n, u64* __restrict__ w, u32* __restrict__ bad) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const u64 v = wit[i];
    if (v >= GLP) atomicOr(bad, 1u);                    // FGL::from(u64) = from_repr(..).unwrap(): not a field element
    w[i] = v;
}
__global__ __launch_bounds__(256) void c12_add_kernel(const AddOp* __restrict__ ops, u64 n, u64* __restrict__ w) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const AddOp o = ops[i];
    w[o.dst] = gl::add(gl::mul(w[o.a], o.ca), gl::mul(w[o.b], o.cb));
}
__global__ __launch_bounds__(256) void c12_gather_kernel(const u64* __restrict__ w, const u32* __restrict__ s_map, u64 n_map, u64 n_total, u64* __restrict__ cm) {
    const u64 i = blockIdx.x * 256ull + threadIdx.x;    // one cell of the [N][12] matrix
    if (i >= n_total) return;
    u64 v = 0;
    if (i < n_map) { const u32 s = s_map[i]; if (s) v = w[s]; }
    cm[i] = v;
}

}  // namespace

struct C12Exec {
    u64 n_witness = 0, adds_len = 0, map_rows = 0;
    std::vector<u64> level_start;     // ops of depth d are [level_start[d], level_start[d + 1])
    DevBuf ops, s_map;
};

C12Exec* c12_exec_new(const char* exec_json, size_t len, uint64_t n_witness) {
    ZK_REQUIRE(exec_json, "compressor12: null exec file");
    const std::vector<u64> v = parse_u64_array(exec_json, len);
    ZK_REQUIRE(v.size() >= 2, "compressor12: exec file too short");
    auto E = std::make_unique<C12Exec>();
    E->n_witness = n_witness; E->adds_len = v[0]; E->map_rows = v[1];
    ZK_REQUIRE(v.size() - 2 == E->adds_len * 4 + E->map_rows * 12, "compressor12: exec file length does not match its header");   // compressor12_exec.rs:117-118
    const u64 n_w = n_witness + E->adds_len;
    ZK_REQUIRE(n_w < (1ull << 32), "compressor12: more than 2^32 wires");
    const u64 rinv = host_powmod(host_mulmod(1ull << 32, 1ull << 32), GLP - 2);   // 2^-64 mod p
    std::vector<u32> depth(E->adds_len);
    std::vector<AddOp> ops(E->adds_len);
    u32 max_depth = 0;
    for (u64 i = 0; i < E->adds_len; ++i) {
        const u64* a = v.data() + 2 + 4 * i;
        ZK_REQUIRE(a[0] < n_witness + i && a[1] < n_witness + i, "compressor12: addition " + std::to_string(i) + " reads a wire that does not exist yet");
        ZK_REQUIRE(a[2] < GLP && a[3] < GLP, "compressor12: coefficient is not a field element");   // from_raw_repr's is_valid
        auto dep = [&](u64 x) { return x < n_witness ? 0u : depth[x - n_witness] + 1; };
        depth[i] = std::max(dep(a[0]), dep(a[1]));
        max_depth = std::max(max_depth, depth[i]);
        ops[i] = AddOp{(u32)a[0], (u32)a[1], (u32)(n_witness + i), 0, host_mulmod(a[2], rinv), host_mulmod(a[3], rinv)};
    }
    // counting sort by depth
    E->level_start.assign((size_t)max_depth + 2, 0);
    for (u64 i = 0; i < E->adds_len; ++i) ++E->level_start[depth[i] + 1];
    for (size_t d = 1; d < E->level_start.size(); ++d) E->level_start[d] += E->level_start[d - 1];
    std::vector<AddOp> sorted(E->adds_len);
    { std::vector<u64> cur(E->level_start.begin(), E->level_start.end() - 1);
      for (u64 i = 0; i < E->adds_len; ++i) sorted[cur[depth[i]]++] = ops[i]; }
    if (E->adds_len == 0) E->level_start.assign(1, 0);
    std::vector<u32> sm(E->map_rows * 12);
    const u64* m = v.data() + 2 + 4 * E->adds_len;
    for (size_t k = 0; k < sm.size(); ++k) {
        ZK_REQUIRE(m[k] < n_w, "compressor12: s_map entry out of range");
        sm[k] = (u32)m[k];
    }
    E->ops.reserve(std::max<size_t>(sorted.size(), 1) * sizeof(AddOp)); E->s_map.reserve(std::max<size_t>(sm.size(), 1) * 4);
    if (!sorted.empty()) h2d_sync(E->ops.p, sorted.data(), sorted.size() * sizeof(AddOp));
    if (!sm.empty()) h2d_sync(E->s_map.p, sm.data(), sm.size() * 4);
    return E.release();
}
void c12_exec_free(C12Exec* e) { delete e; }
uint64_t c12_exec_levels(const C12Exec* e) { return e->level_start.size() - 1; }

// d_witness: n_witness canonical u64 (a value >= p is an error, as FGL::from's unwrap); d_cm: [n_rows][12] words, n_rows >= s_map_column_len
void c12_exec_dev(const C12Exec* E, const u64* d_witness, uint64_t n_witness, uint64_t n_rows, u64* d_cm, hipStream_t st) {
    ZK_REQUIRE(E && d_witness && d_cm, "compressor12: null argument");
    ZK_REQUIRE(n_witness == E->n_witness, "compressor12: the witness has " + std::to_string(n_witness) + " values, the circuit " + std::to_string(E->n_witness));
    ZK_REQUIRE(n_rows >= E->map_rows, "compressor12: s_map has more rows than the trace");
    DevBuf w;
    const u64 n_w = E->n_witness + E->adds_len;
    w.reserve((n_w + 1) * 8);                             // + one word for the validity flag
    u32* d_bad = (u32*)(w.u() + n_w);
    ZK_HIP(hipMemsetAsync(d_bad, 0, 8, st));
    auto blocks = [](u64 n) { return dim3((unsigned)((n + 255) / 256)); };
    if (n_witness) hipLaunchKernelGGL(c12_load_kernel, blocks(n_witness), dim3(256), 0, st, d_witness, n_witness, w.u(), d_bad);
    for (size_t d = 0; d + 1 < E->level_start.size(); ++d) {
        const u64 lo = E->level_start[d], cnt = E->level_start[d + 1] - lo;
        if (cnt) hipLaunchKernelGGL(c12_add_kernel, blocks(cnt), dim3(256), 0, st, (const AddOp*)E->ops.p + lo, cnt, w.u());
    }
    if (n_rows) hipLaunchKernelGGL(c12_gather_kernel, blocks(n_rows * 12), dim3(256), 0, st, (const u64*)w.u(), (const u32*)E->s_map.p, E->map_rows * 12, n_rows * 12, d_cm);
    ZK_HIP(hipGetLastError());
    u32 bad = 0;
    ZK_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));   // w goes back to the pool
    ZK_REQUIRE(!bad, "compressor12: a witness value is not a Goldilocks field element");
}

}  // namespace zk
