// C ABI of libzkgpu (include/zkgpu.h): thin extern "C" wrappers, error capture, host<->device
// staging for the host-pointer entry points, and the Merkle tree handle.
#include "zk_internal.h"
#include "../../include/zkgpu.h"
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
#include <cstring>

namespace zk {

static thread_local std::string t_err;
void set_error(const std::string& msg) { t_err = msg; }

// ---- caching device allocator -------------------------------------------------------------------
namespace {
// An event recorded when blocks were freed; shared by all blocks freed in one burst (pool_defer_*), recycled by the last of them
struct Ev { hipEvent_t e = nullptr; int refs = 0; };
struct Block { size_t bytes = 0; std::vector<Ev*> pending; };
std::mutex g_pool_mu;
std::multimap<size_t, void*> g_pool_free;   // size -> idle block
std::map<void*, Block> g_pool_blocks;       // every block handed out by pool_alloc
std::vector<hipStream_t> g_streams{nullptr};  // streams the library has been asked to work on
std::vector<hipEvent_t> g_event_cache;
thread_local hipStream_t t_stream = nullptr;
thread_local std::vector<hipStream_t> t_streams;   // non-null streams this host thread has issued on
thread_local int t_depth = 0;                      // C-ABI calls in progress on this thread (the prover calls entry points itself)
thread_local bool t_defer = false;                 // pool_defer_begin(): frees are collected ...
thread_local std::vector<void*> t_deferred;        // ... here, and stamped with ONE set of events by pool_defer_flush()
void ev_release(Ev* v) { if (--v->refs == 0) { g_event_cache.push_back(v->e); delete v; } }   // g_pool_mu held

hipEvent_t event_get() {                     // g_pool_mu held; never throws (DevBuf destructors end up here): nullptr = no event to be had
    if (!g_event_cache.empty()) { hipEvent_t e = g_event_cache.back(); g_event_cache.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
}
hipStream_t cur_stream() { return t_stream; }
// The GPU of the process (zk_init).  HIP's current device is a property of the host thread and every new thread starts on device 0,
// so a prover thread of rank k > 0 would otherwise allocate and launch on GPU 0: each thread is bound on its first call.
static std::atomic<int> g_device{-1};          // -1: zk_init was never called, threads are left as the caller set them up
// zk_init's device overrides whatever the caller (torch.cuda.set_device, another library) made current on this thread in the
// meantime: pooled blocks, constant tables and code modules all belong to device d, so every outermost call re-checks the thread's
// actual device (hipGetDevice is a thread-local read) instead of trusting a flag cached at the first call.
void bind_device() noexcept {
    const int d = g_device.load(std::memory_order_relaxed);
    if (d < 0) return;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == d) return;
    if (hipSetDevice(d) != hipSuccess) (void)hipGetLastError();   // the work that follows reports its own error
}
CallScope::CallScope() : saved(t_stream) { if (t_depth++ == 0) { t_stream = nullptr; bind_device(); } }   // a call from outside starts on the null stream,
CallScope::~CallScope() { --t_depth; t_stream = saved; }                                // one made by the prover inherits the prover's
hipStream_t on_stream(hipStream_t st) {
    t_stream = st;
    if (st) {
        bool mine = false;
        for (hipStream_t s : t_streams) mine |= s == st;
        if (!mine) {
            t_streams.push_back(st);
            std::lock_guard<std::mutex> lk(g_pool_mu);
            bool known = false;
            for (hipStream_t s : g_streams) known |= s == st;
            if (!known) {
                // While the null stream was the only one in use, blocks went back to the pool without an event (pool_free):
                // work queued there may still be running on them, and a non-blocking stream does not wait for the null stream.
                // Drain once, at the moment a second stream appears; from here on every free records its events.
                // (under the lock on purpose: no block may be handed out between the drain and the registration)
                if (g_streams.size() == 1) (void)hipDeviceSynchronize();
                g_streams.push_back(st);
            }
        }
    }
    return st;
}
// A helper stream that lives inside one call (the MSM's sort stream): registered so that this thread's frees are stamped on it while it
// is in use, WITHOUT the one-off device drain of on_stream -- the caller guarantees that the side stream's first operation waits for an
// event recorded on the current stream after every buffer it will touch was allocated (so it is ordered behind those blocks' previous
// users), and calls forget_stream once the current stream has waited for the side stream's last event.  The thread's current stream is kept.
void on_side_stream(hipStream_t ss) {
    if (!ss) return;
    bool mine = false;
    for (hipStream_t s : t_streams) mine |= s == ss;
    if (!mine) t_streams.push_back(ss);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    bool known = false;
    for (hipStream_t s : g_streams) known |= s == ss;
    if (!known) g_streams.push_back(ss);
}
void* pool_alloc(size_t bytes, bool host_wait) {
    if (bytes == 0) bytes = 8;
    {
        std::unique_lock<std::mutex> lk(g_pool_mu);
        auto it = g_pool_free.find(bytes);
        if (it != g_pool_free.end()) {
            void* p = it->second; g_pool_free.erase(it);
            std::vector<Ev*> pending;
            pending.swap(g_pool_blocks[p].pending);                   // the block is out of the free list: nobody else sees it or its events' list
            // whoever used the block last finishes first.  The library's own buffers: the stream this thread works on waits
            // (asynchronous, under the lock).  A block that leaves the library (zk_dev_alloc: the caller may touch it from any
            // stream): the HOST waits -- with the lock released, so that other provers' allocations and frees go on meanwhile.
            if (host_wait) lk.unlock();
            for (Ev* v : pending) {
                const hipError_t rc = host_wait ? hipEventSynchronize(v->e) : hipStreamWaitEvent(t_stream, v->e, 0);
                // an event whose stream has been destroyed since (a released setup's side stream: drained before it went) reports an
                // error here; its work is done, and the error must not surface at some later hipGetLastError()
                if (rc != hipSuccess) (void)hipGetLastError();
            }
            if (host_wait) lk.lock();
            for (Ev* v : pending) ev_release(v);
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // out of memory: drop the cache and retry once
        (void)hipGetLastError();                                      // the failed attempt must not surface at a later check
        pool_trim();
        ZK_HIP(hipMalloc(&p, bytes));
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_blocks[p].bytes = bytes;
    return p;
}
namespace {
// events for blocks freed now: the null stream and every stream the FREEING thread has issued on (a buffer is released by the
// thread that owns it; work another thread did on it was ordered before this thread's by whoever handed it over).  Streams of
// other threads are left alone: concurrent provers must not wait for each other.  g_pool_mu held.
std::vector<Ev*> stamp_now() {
    std::vector<Ev*> out;
    if (g_streams.size() <= 1) return out;               // a single stream in use: reuse is stream ordered (on_stream drains at the second)
    auto record = [&](hipStream_t st) {
        hipEvent_t e = event_get();
        if (!e) { (void)hipStreamSynchronize(st); (void)hipGetLastError(); return; }   // no event to be had: wait here instead
        if (hipEventRecord(e, st) == hipSuccess) out.push_back(new Ev{e, 0});
        else { (void)hipGetLastError(); g_event_cache.push_back(e); }   // a destroyed stream has nothing in flight
    };
    record(nullptr);
    for (hipStream_t st : t_streams) record(st);
    return out;
}
void release_stamped(void* p, const std::vector<Ev*>& evs) {   // g_pool_mu held
    auto it = g_pool_blocks.find(p);
    if (it == g_pool_blocks.end()) { (void)hipFree(p); return; }
    for (Ev* v : evs) { ++v->refs; it->second.pending.push_back(v); }
    g_pool_free.emplace(it->second.bytes, p);
}
}  // namespace
void pool_free(void* p) {
    if (!p) return;
    if (t_defer) { t_deferred.push_back(p); return; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const std::vector<Ev*> evs = stamp_now();
    release_stamped(p, evs);
    for (Ev* v : evs) if (v->refs == 0) { g_event_cache.push_back(v->e); delete v; }   // the block was not the pool's
}
// A burst of frees with no launch in between (the end of a proof: ~70 buffers and trees go at once) shares one set of events
// instead of recording two per block: pool_defer_begin() after the last launch, pool_defer_flush() once the destructors have run.
void pool_defer_begin() { t_defer = true; }
void pool_defer_flush() {
    t_defer = false;
    if (t_deferred.empty()) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const std::vector<Ev*> evs = stamp_now();
    for (void* p : t_deferred) release_stamped(p, evs);
    for (Ev* v : evs) if (v->refs == 0) { g_event_cache.push_back(v->e); delete v; }
    t_deferred.clear();
}
// Host <-> device copies of the library's own pooled buffers: on the stream this thread is working on (the one whose
// queue was ordered behind the buffer's previous user by pool_alloc), then waited for -- a plain hipMemcpy runs on the
// null stream, which a non-blocking stream does not synchronise with.
void h2d_sync(void* d, const void* h, size_t n) {
    if (!n) return;
    ZK_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, t_stream));
    ZK_HIP(hipStreamSynchronize(t_stream));
}
void d2h_sync(void* h, const void* d, size_t n) {
    if (!n) return;
    ZK_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, t_stream));
    ZK_HIP(hipStreamSynchronize(t_stream));
}
void pool_trim() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& kv : g_pool_free) {
        Block& b = g_pool_blocks[kv.second];
        for (Ev* v : b.pending) { if (hipEventSynchronize(v->e) != hipSuccess) (void)hipGetLastError(); ev_release(v); }
        (void)hipFree(kv.second); g_pool_blocks.erase(kv.second);
    }
    g_pool_free.clear();
}
void forget_stream(hipStream_t st) {                                  // before hipStreamDestroy
    for (size_t i = 0; i < t_streams.size(); ++i)
        if (t_streams[i] == st) { t_streams.erase(t_streams.begin() + i); break; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 1; i < g_streams.size(); ++i)
        if (g_streams[i] == st) { g_streams.erase(g_streams.begin() + i); break; }
}

namespace {

std::mutex g_ws_mu;
// grow-only staging for the host-pointer API.  Never destroyed: a static DevBuf would go back to the pool from a static
// destructor, after the main thread's thread_local stream list is gone and possibly after the HIP runtime's own teardown
// (a process that still had a registered side stream at exit crashed there).
DevBuf &g_ws_a = *new DevBuf, &g_ws_b = *new DevBuf, &g_ws_c = *new DevBuf;

template <class F>
int guard(F&& f) {
    CallScope scope;
    try { f(); return 0; }
    catch (const std::exception& e) { set_error(e.what()); return -1; }
    catch (...) { set_error("unknown error"); return -1; }
}

__global__ void gather_proof_kernel(const u64* __restrict__ elements, const u64* __restrict__ nodes, u32 width,
                                    u64 height, u64 idx, u64* __restrict__ out /* width + depth*4 */) {
    const u32 t = threadIdx.x;
    for (u32 i = t; i < width; i += blockDim.x) out[i] = elements[idx * width + i];
    if (t == 0) {  // merklehash.rs:64-76 merkle_gen_merkle_proof
        u64 n = height, off = 0, id = idx; u32 d = 0;
        while (n > 1) {
            const u64* sib = nodes + 4 * (off + (id ^ 1));
            for (int k = 0; k < 4; ++k) out[width + 4 * d + k] = sib[k];
            u64 next = (n - 1) / 2 + 1;
            off += next * 2; n = next; id >>= 1; ++d;
        }
    }
}

// synthetic words for benchmarks and size tests: word i = splitmix64(seed + i) folded below p (one conditional subtraction)
__global__ void fill_splitmix_kernel(u64* __restrict__ out, u64 n, u64 seed) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = seed + i + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    out[i] = z >= GL_P ? z - GL_P : z;
}

// the same for n queries at once: block q serves idx[q], out + q * (width + 4 * depth)
__global__ void gather_proofs_kernel(const u64* __restrict__ elements, const u64* __restrict__ nodes, u32 width, u64 height,
                                     u32 depth, const u64* __restrict__ idxs, u64* __restrict__ outs, u64 mask) {
    const u32 t = threadIdx.x;
    const u64 idx = idxs[blockIdx.x] & mask;             // mask: the query index reduced to a later FRI step's domain (fri.rs:166-168)
    u64* __restrict__ out = outs + (u64)blockIdx.x * (width + 4 * depth);
    for (u32 i = t; i < width; i += blockDim.x) out[i] = elements[idx * width + i];
    if (t == 0) {
        u64 n = height, off = 0, id = idx; u32 d = 0;
        while (n > 1) {
            const u64* sib = nodes + 4 * (off + (id ^ 1));
            for (int k = 0; k < 4; ++k) out[width + 4 * d + k] = sib[k];
            u64 next = (n - 1) / 2 + 1;
            off += next * 2; n = next; id >>= 1; ++d;
        }
    }
}
// every tree of a proof in ONE launch: block (q, j) serves query q of tree j (a small proof opened its seven or eight trees in as many
// launches of ~5 us each); lane d copies the sibling of level d
struct GatherMulti { const u64* elements[16]; const u64* nodes[16]; u64* out[16]; u64 height[16], mask[16]; u32 width[16], depth[16]; };
__global__ void gather_proofs_multi_kernel(const GatherMulti G, const u64* __restrict__ idxs) {
    const u32 t = threadIdx.x, j = blockIdx.y;
    const u32 width = G.width[j], depth = G.depth[j];
    const u64 idx = idxs[blockIdx.x] & G.mask[j];
    const u64* __restrict__ elements = G.elements[j];
    const u64* __restrict__ nodes = G.nodes[j];
    u64* __restrict__ out = G.out[j] + (u64)blockIdx.x * (width + 4 * depth);
    for (u32 i = t; i < width; i += blockDim.x) out[i] = elements[idx * width + i];
    for (u32 d = t; d < depth; d += blockDim.x) {
        u64 n = G.height[j], off = 0;
        for (u32 k = 0; k < d; ++k) { const u64 next = (n - 1) / 2 + 1; off += next * 2; n = next; }
        const u64* sib = nodes + 4 * (off + ((idx >> d) ^ 1));
        for (int k = 0; k < 4; ++k) out[width + 4 * d + k] = sib[k];
    }
}
// arity-16 trees of the scalar-field hashes (merklehash_bn128.rs:86-106): block q -> row + depth groups of 16 digests
__global__ void fr_gather_proofs_kernel(const u64* __restrict__ elements, const u64* __restrict__ nodes, u32 width, u64 height,
                                        u32 depth, const u64* __restrict__ idxs, u64* __restrict__ outs) {
    const u32 t = threadIdx.x;
    const u64 idx = idxs[blockIdx.x];
    u64* __restrict__ out = outs + (u64)blockIdx.x * (width + 64 * depth);
    for (u32 i = t; i < width; i += blockDim.x) out[i] = elements[idx * width + i];
    u64 n = height, off = 0, id = idx; u32 d = 0;
    while (n > 1) {
        const u64 si = id & ~(u64)15;
        if (t < 64) out[width + 64 * d + t] = nodes[4 * (off + si) + t];
        const u64 next = (n - 1) / 16 + 1;
        off += next * 16; n = next; id >>= 4; ++d;
    }
}

}  // namespace
}  // namespace zk

using namespace zk;

struct zk_merkle {
    const u64* d_elements = nullptr;  // [height][width]
    DevBuf owned_elements;            // set when the tree owns its rows
    DevBuf nodes;                     // merkle_n_nodes(height) * 4 words
    DevBuf proof;                     // staging for group proofs
    uint32_t width = 0, depth = 0;
    uint64_t height = 0, n_nodes = 0;
    hipStream_t stream = nullptr;
};

struct zk_transcript {
    DevBuf state;  // TranscriptState (poseidon.hip)
    DevBuf io;     // staging for host-word put / get
    hipStream_t stream = nullptr;  // where the state was last worked on: the host-word calls continue (and wait) there
};

static uint32_t tree_depth(uint64_t height) {
    uint32_t d = 0; uint64_t n = height;
    while (n > 1) { n = (n - 1) / 2 + 1; ++d; }
    return d;
}

extern "C" {

int zk_init(int device) {
    return guard([&] {
        int n = 0;
        ZK_HIP(hipGetDeviceCount(&n));
        ZK_REQUIRE(device >= 0 && device < n, "zk_init: no such device");
        ZK_HIP(hipSetDevice(device));
        g_device.store(device, std::memory_order_relaxed);   // ... and of every thread that calls into the library from now on
    });
}
const char* zk_last_error(void) { return t_err.c_str(); }
int zk_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
uint64_t zk_gl_modulus(void) { return GL_P; }
uint64_t zk_gl_root_of_unity(uint32_t k) { return k <= 32 ? gl::hroot(k) : 0; }

void* zk_dev_alloc(size_t bytes) {
    void* p = nullptr;
    // handed to the caller, who may write it on any stream: reuse is ordered on the host, not on the null stream of this call
    if (guard([&] { p = pool_alloc(bytes, /*host_wait=*/true); }) != 0) return nullptr;
    return p;
}
int zk_dev_free(void* p) { return guard([&] { pool_free(p); }); }
int zk_dev_trim(void) { return guard([&] { ZK_HIP(hipDeviceSynchronize()); pool_trim(); }); }
int zk_dev_upload(void* d, const void* h, size_t n) { return guard([&] { ZK_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); }); }
int zk_dev_download(void* h, const void* d, size_t n) { return guard([&] { ZK_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); }); }
int zk_dev_sync(void) { return guard([&] { ZK_HIP(hipDeviceSynchronize()); }); }
void* zk_stream_new(void) {
    hipStream_t st = nullptr;
    if (guard([&] { ZK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }) != 0) return nullptr;
    return st;
}
int zk_stream_sync(void* stream) { return guard([&] { ZK_HIP(hipStreamSynchronize((hipStream_t)stream)); }); }
int zk_stream_free(void* stream) {
    return guard([&] {
        if (!stream) return;
        ZK_HIP(hipStreamSynchronize((hipStream_t)stream));
        forget_stream((hipStream_t)stream);
        ZK_HIP(hipStreamDestroy((hipStream_t)stream));
    });
}
int zk_dev_fill_splitmix(uint64_t* d, uint64_t n_words, uint64_t seed, void* stream) {
    return guard([&] {
        if (!n_words) return;
        ZK_REQUIRE(d != nullptr, "zk_dev_fill_splitmix: null buffer");
        ZK_REQUIRE((n_words + 255) / 256 < (1ull << 31), "zk_dev_fill_splitmix: too many words for one launch");
        hipLaunchKernelGGL(fill_splitmix_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, on_stream((hipStream_t)stream), (u64*)d, n_words, seed);
        ZK_HIP(hipGetLastError());
    });
}
int zk_dev_memset(void* d, int value, size_t n) { return guard([&] { if (n) ZK_HIP(hipMemset(d, value, n)); }); }

int zk_gl_ntt_passes(uint32_t nbits) { return ntt_num_passes(nbits); }

int zk_gl_ntt_dev(const uint64_t* d_src, uint64_t* d_dst, uint64_t* d_tmp, uint32_t n_pols, uint32_t nbits,
                  int inverse, void* stream) {
    return guard([&] {
        ZK_REQUIRE(nbits <= 32, "zk_gl_ntt: nbits > 32");
        ZK_REQUIRE(d_tmp != nullptr || ntt_num_passes(nbits) == 1 || n_pols == 0, "zk_gl_ntt_dev: d_tmp required");
        ntt_dev((const u64*)d_src, (u64*)d_dst, (u64*)d_tmp, n_pols, nbits, inverse != 0, on_stream((hipStream_t)stream));
    });
}

int zk_gl_lde_dev(const uint64_t* d_src, uint32_t n_pols, uint32_t nbits, uint64_t* d_dst, uint64_t* d_tmp,
                  uint32_t nbits_ext, void* stream) {
    return guard([&] {
        ZK_REQUIRE(d_tmp != nullptr || n_pols == 0, "zk_gl_lde_dev: d_tmp required");
        lde_dev((const u64*)d_src, (u64*)d_dst, (u64*)d_tmp, n_pols, nbits, nbits_ext, on_stream((hipStream_t)stream));
    });
}

int zk_gl_ntt(const uint64_t* src, uint64_t* dst, uint32_t n_pols, uint32_t nbits, int inverse) {
    return guard([&] {
        ZK_REQUIRE(nbits <= 32, "zk_gl_ntt: nbits > 32");
        if (n_pols == 0) return;
        ZK_REQUIRE(src && dst && src != dst, "zk_gl_ntt: bad buffers (dst may not alias src)");
        std::lock_guard<std::mutex> lk(g_ws_mu);
        const size_t bytes = ((size_t)1 << nbits) * n_pols * sizeof(u64);
        g_ws_a.reserve(bytes); g_ws_b.reserve(bytes); g_ws_c.reserve(bytes);
        ZK_HIP(hipMemcpy(g_ws_a.p, src, bytes, hipMemcpyHostToDevice));
        ntt_dev(g_ws_a.u(), g_ws_b.u(), g_ws_c.u(), n_pols, nbits, inverse != 0, nullptr);
        ZK_HIP(hipMemcpy(dst, g_ws_b.p, bytes, hipMemcpyDeviceToHost));
    });
}

int zk_gl_lde(const uint64_t* src, uint32_t n_pols, uint32_t nbits, uint64_t* dst, uint32_t nbits_ext) {
    return guard([&] {
        ZK_REQUIRE(nbits_ext <= 32 && nbits <= nbits_ext, "zk_gl_lde: need nbits <= nbits_ext <= 32");
        if (n_pols == 0) return;  // fft_p.rs:262-264: empty source is a no-op
        ZK_REQUIRE(src && dst, "zk_gl_lde: null buffer");
        std::lock_guard<std::mutex> lk(g_ws_mu);
        const size_t in_bytes = ((size_t)1 << nbits) * n_pols * sizeof(u64);
        const size_t out_bytes = ((size_t)1 << nbits_ext) * n_pols * sizeof(u64);
        g_ws_a.reserve(in_bytes); g_ws_b.reserve(out_bytes); g_ws_c.reserve(out_bytes);
        ZK_HIP(hipMemcpy(g_ws_a.p, src, in_bytes, hipMemcpyHostToDevice));
        lde_dev(g_ws_a.u(), g_ws_b.u(), g_ws_c.u(), n_pols, nbits, nbits_ext, nullptr);
        ZK_HIP(hipMemcpy(dst, g_ws_b.p, out_bytes, hipMemcpyDeviceToHost));
    });
}

int zk_gl_poseidon_selfcheck(void) {                                   // host arithmetic only: no CallScope, no device
    try {
        const std::string why = poseidon_tables_selfcheck();
        if (why.empty()) return 0;
        set_error(why);
    } catch (const std::exception& e) { set_error(e.what()); }
    return -1;
}

int zk_gl_poseidon(const uint64_t in[8], const uint64_t cap[4], uint64_t* out, uint32_t n_out) {
    return guard([&] {
        // poseidon_opt.rs:81-96 rejects wrong input/capacity lengths; with fixed-size C arrays the
        // remaining argument error is the output count.
        ZK_REQUIRE(in && cap && out, "zk_gl_poseidon: null buffer");
        ZK_REQUIRE(n_out >= 1 && n_out <= 12, "zk_gl_poseidon: n_out must be 1..12");
        std::lock_guard<std::mutex> lk(g_ws_mu);
        g_ws_a.reserve(24 * sizeof(u64));
        u64 h[12];
        memcpy(h, in, 64); memcpy(h + 8, cap, 32);
        ZK_HIP(hipMemcpy(g_ws_a.p, h, 96, hipMemcpyHostToDevice));
        poseidon_dev(g_ws_a.u(), g_ws_a.u() + 8, g_ws_a.u() + 12, (int)n_out, nullptr);
        ZK_HIP(hipMemcpy(out, g_ws_a.u() + 12, n_out * sizeof(u64), hipMemcpyDeviceToHost));
    });
}

int zk_gl_linearhash_rows_dev(const uint64_t* d_rows, uint32_t width, uint64_t height, uint64_t* d_digests, void* stream) {
    return guard([&] { linearhash_rows_dev((const u64*)d_rows, width, height, (u64*)d_digests, on_stream((hipStream_t)stream)); });
}

int zk_gl_linearhash(const uint64_t* v, size_t n, uint64_t out[4]) {
    return guard([&] {
        ZK_REQUIRE(out && (v || n == 0), "zk_gl_linearhash: null buffer");
        ZK_REQUIRE(n < (1ull << 32), "zk_gl_linearhash: row too wide");
        std::lock_guard<std::mutex> lk(g_ws_mu);
        g_ws_a.reserve((n + 8) * sizeof(u64));
        if (n) ZK_HIP(hipMemcpy(g_ws_a.p, v, n * sizeof(u64), hipMemcpyHostToDevice));
        u64* d_out = g_ws_a.u() + n;
        linearhash_rows_dev(g_ws_a.u(), (uint32_t)n, 1, d_out, nullptr);
        ZK_HIP(hipMemcpy(out, d_out, 32, hipMemcpyDeviceToHost));
    });
}

uint64_t zk_merkle_n_nodes(uint64_t height) { return height ? merkle_n_nodes(height) : 0; }

static zk_merkle_t* make_tree(const u64* d_rows, zk_merkle_t* t, uint32_t width, uint64_t height, hipStream_t st) {
    t->d_elements = d_rows; t->width = width; t->height = height;
    t->n_nodes = merkle_n_nodes(height); t->depth = tree_depth(height); t->stream = st;
    t->nodes.reserve(t->n_nodes * 32);
    t->proof.reserve(((size_t)width + 4 * (size_t)t->depth + 4) * sizeof(u64));
    merkelize_dev(d_rows, width, height, t->nodes.u(), st);
    return t;
}

zk_merkle_t* zk_gl_merkelize(const uint64_t* buff, uint32_t width, uint64_t height) {
    zk_merkle_t* t = nullptr;
    int rc = guard([&] {
        ZK_REQUIRE((buff || width == 0) && height >= 1, "zk_gl_merkelize: empty matrix");
        t = new zk_merkle();
        const size_t bytes = (size_t)width * height * sizeof(u64);
        t->owned_elements.reserve(bytes ? bytes : 8);
        if (bytes) ZK_HIP(hipMemcpy(t->owned_elements.p, buff, bytes, hipMemcpyHostToDevice));
        make_tree(t->owned_elements.u(), t, width, height, nullptr);
        ZK_HIP(hipStreamSynchronize(nullptr));
    });
    if (rc != 0) { delete t; return nullptr; }
    return t;
}

zk_merkle_t* zk_gl_merkelize_dev(const uint64_t* d_buff, uint32_t width, uint64_t height, void* stream) {
    zk_merkle_t* t = nullptr;
    int rc = guard([&] {
        ZK_REQUIRE((d_buff || width == 0) && height >= 1, "zk_gl_merkelize_dev: empty matrix");
        t = new zk_merkle();
        make_tree((const u64*)d_buff, t, width, height, on_stream((hipStream_t)stream));
    });
    if (rc != 0) { delete t; return nullptr; }
    return t;
}

int zk_merkle_root(const zk_merkle_t* t, uint64_t out[4]) {
    return guard([&] {
        ZK_REQUIRE(t && out, "zk_merkle_root: null");
        ZK_HIP(hipStreamSynchronize(t->stream));
        ZK_HIP(hipMemcpy(out, t->nodes.u() + 4 * (t->n_nodes - 1), 32, hipMemcpyDeviceToHost));
    });
}

int zk_merkle_nodes(const zk_merkle_t* t, uint64_t* out) {
    return guard([&] {
        ZK_REQUIRE(t && out, "zk_merkle_nodes: null");
        ZK_HIP(hipStreamSynchronize(t->stream));
        ZK_HIP(hipMemcpy(out, t->nodes.p, t->n_nodes * 32, hipMemcpyDeviceToHost));
    });
}

int zk_merkle_elements(const zk_merkle_t* t, uint64_t* out) {
    return guard([&] {
        ZK_REQUIRE(t && out, "zk_merkle_elements: null");
        ZK_HIP(hipStreamSynchronize(t->stream));
        ZK_HIP(hipMemcpy(out, t->d_elements, (size_t)t->height * t->width * 8, hipMemcpyDeviceToHost));
    });
}

uint32_t zk_merkle_depth(const zk_merkle_t* t) { return t ? t->depth : 0; }
}  // extern "C"
namespace zk {
uint32_t merkle_width(const zk_merkle* t) { return t->width; }
uint64_t merkle_height(const zk_merkle* t) { return t->height; }
}
extern "C" {

int zk_merkle_group_proof(const zk_merkle_t* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out) {
    return guard([&] {
        ZK_REQUIRE(t && row_out && (path_out || t->depth == 0), "zk_merkle_group_proof: null");
        ZK_REQUIRE(idx < t->height, "MerkleTreeError: access invalid node");  // merklehash.rs:431-433
        hipLaunchKernelGGL(gather_proof_kernel, dim3(1), dim3(64), 0, t->stream, t->d_elements, t->nodes.u(),
                           t->width, t->height, idx, t->proof.u());
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(t->stream));
        ZK_HIP(hipMemcpy(row_out, t->proof.p, t->width * sizeof(u64), hipMemcpyDeviceToHost));
        if (t->depth)
            ZK_HIP(hipMemcpy(path_out, t->proof.u() + t->width, (size_t)t->depth * 32, hipMemcpyDeviceToHost));
    });
}

int zk_merkle_group_proofs(const zk_merkle_t* t, const uint64_t* idx, uint32_t n, uint64_t* rows_out, uint64_t* paths_out) {
    return guard([&] {
        ZK_REQUIRE(t && (n == 0 || (idx && rows_out && (paths_out || t->depth == 0))), "zk_merkle_group_proofs: null");
        if (n == 0) return;
        for (uint32_t q = 0; q < n; ++q) ZK_REQUIRE(idx[q] < t->height, "MerkleTreeError: access invalid node");
        const size_t per = (size_t)t->width + 4 * (size_t)t->depth;
        on_stream(t->stream);
        DevBuf d_idx, d_out; d_idx.reserve(n * 8); d_out.reserve(std::max<size_t>(1, per * n) * 8);
        ZK_HIP(hipMemcpyAsync(d_idx.p, idx, n * 8, hipMemcpyHostToDevice, t->stream));
        hipLaunchKernelGGL(gather_proofs_kernel, dim3(n), dim3(64), 0, t->stream, t->d_elements, t->nodes.u(), t->width, t->height,
                           t->depth, d_idx.u(), d_out.u(), ~0ull);
        ZK_HIP(hipGetLastError());
        std::vector<u64> h(std::max<size_t>(1, per * n));
        ZK_HIP(hipMemcpyAsync(h.data(), d_out.p, per * n * 8, hipMemcpyDeviceToHost, t->stream));
        ZK_HIP(hipStreamSynchronize(t->stream));
        for (uint32_t q = 0; q < n; ++q) {
            memcpy(rows_out + (size_t)q * t->width, h.data() + q * per, (size_t)t->width * 8);
            if (t->depth) memcpy(paths_out + (size_t)q * 4 * t->depth, h.data() + q * per + t->width, (size_t)t->depth * 32);
        }
    });
}

}  // extern "C"
namespace zk {
void merkle_group_proofs_async(const zk_merkle* t, const u64* d_idx, uint32_t n, u64* d_out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(gather_proofs_kernel, dim3(n), dim3(64), 0, st, t->d_elements, t->nodes.u(), t->width, t->height, t->depth, d_idx, d_out, ~0ull);
    ZK_HIP(hipGetLastError());
}
// the same at the indices d_idx[q] & mask (mask + 1 = the tree's height, a power of two): the query indices stay on the device
void merkle_group_proofs_masked_async(const zk_merkle* t, const u64* d_idx, u64 mask, uint32_t n, u64* d_out, hipStream_t st) {
    if (n == 0) return;
    ZK_REQUIRE(mask < t->height, "MerkleTreeError: access invalid node");
    hipLaunchKernelGGL(gather_proofs_kernel, dim3(n), dim3(64), 0, st, t->d_elements, t->nodes.u(), t->width, t->height, t->depth, d_idx, d_out, mask);
    ZK_HIP(hipGetLastError());
}
// up to 16 trees at once: tree j at d_idx[q] & mask[j] into d_out[j] (rows of width_j + 4 depth_j words per query)
void merkle_group_proofs_multi_async(const zk_merkle* const* trees, const u64* masks, u64* const* d_outs, uint32_t n_trees, const u64* d_idx, uint32_t n, hipStream_t st) {
    if (n == 0 || n_trees == 0) return;
    for (uint32_t j0 = 0; j0 < n_trees; j0 += 16) {
        GatherMulti G; memset(&G, 0, sizeof G);
        const uint32_t m = std::min<uint32_t>(16, n_trees - j0);
        for (uint32_t j = 0; j < m; ++j) {
            const zk_merkle* t = trees[j0 + j];
            ZK_REQUIRE(masks[j0 + j] < t->height, "MerkleTreeError: access invalid node");
            G.elements[j] = t->d_elements; G.nodes[j] = t->nodes.u(); G.out[j] = d_outs[j0 + j];
            G.height[j] = t->height; G.mask[j] = masks[j0 + j]; G.width[j] = t->width; G.depth[j] = t->depth;
        }
        hipLaunchKernelGGL(gather_proofs_multi_kernel, dim3(n, m), dim3(64), 0, st, G, d_idx);
        ZK_HIP(hipGetLastError());
    }
}
// get_permutations (transcript.rs:73-102) into device memory, on `st`: the indices feed the gather kernels without visiting the host
void transcript_permutations_async(zk_transcript* t, uint32_t n, uint32_t nbits, u64* d_dst, hipStream_t st) {
    t->stream = st;
    transcript_permutations_dev(t->state.p, n, nbits, d_dst, st);
}
void transcript_put_get_async(zk_transcript* t, const u64* d_src, uint64_t n_put, u64* d_dst, uint32_t n_get, uint32_t bits, hipStream_t st) {
    if (t->stream != st) { ZK_HIP(hipStreamSynchronize(t->stream)); t->stream = st; }
    transcript_put_get_dev(t->state.p, d_src, n_put, d_dst, n_get, bits, st);
}
}  // namespace zk
extern "C" {

// ---- transcript ------------------------------------------------------------------------------
// the sponge state lives on one stream at a time: work moves to `st` after whatever was issued on the previous stream
static hipStream_t transcript_stream(zk_transcript_t* t, hipStream_t st) {
    if (t->stream != st) { ZK_HIP(hipStreamSynchronize(t->stream)); t->stream = st; }
    return on_stream(st);
}
zk_transcript_t* zk_transcript_new(void) {
    zk_transcript_t* t = nullptr;
    int rc = guard([&] {
        t = new zk_transcript();
        t->stream = cur_stream();                                       // the caller's (a prover's own stream, or the null stream)
        t->state.reserve(transcript_state_bytes());
        t->io.reserve(4096 * sizeof(u64));
        transcript_init_dev(t->state.p, t->stream);
    });
    if (rc != 0) { delete t; return nullptr; }
    return t;
}
int zk_transcript_put_dev(zk_transcript_t* t, const uint64_t* d_src, size_t n, void* stream) {
    return guard([&] { ZK_REQUIRE(t, "transcript: null"); transcript_put_dev(t->state.p, (const u64*)d_src, n, transcript_stream(t, (hipStream_t)stream)); });
}
int zk_transcript_put(zk_transcript_t* t, const uint64_t* src, size_t n) {
    return guard([&] {
        ZK_REQUIRE(t && (src || n == 0), "transcript: null");
        if (n == 0) return;
        on_stream(t->stream);
        t->io.reserve(n * sizeof(u64));
        ZK_HIP(hipStreamSynchronize(t->stream));                         // io may still be read by an earlier put
        ZK_HIP(hipMemcpy(t->io.p, src, n * sizeof(u64), hipMemcpyHostToDevice));
        transcript_put_dev(t->state.p, t->io.u(), n, on_stream(t->stream));
        ZK_HIP(hipStreamSynchronize(t->stream));
    });
}
int zk_transcript_get_field_dev(zk_transcript_t* t, uint64_t* d_out3, void* stream) {
    return guard([&] { ZK_REQUIRE(t && d_out3, "transcript: null"); transcript_get_dev(t->state.p, (u64*)d_out3, 3, transcript_stream(t, (hipStream_t)stream)); });
}
static int transcript_get_host(zk_transcript_t* t, uint64_t* out, uint32_t n_words) {
    return guard([&] {
        ZK_REQUIRE(t && out, "transcript: null");
        transcript_get_dev(t->state.p, t->io.u(), n_words, on_stream(t->stream));
        ZK_HIP(hipStreamSynchronize(t->stream));
        ZK_HIP(hipMemcpy(out, t->io.p, n_words * sizeof(u64), hipMemcpyDeviceToHost));
    });
}
int zk_transcript_get_field(zk_transcript_t* t, uint64_t out[3]) { return transcript_get_host(t, out, 3); }
int zk_transcript_get_fields1(zk_transcript_t* t, uint64_t* out) { return transcript_get_host(t, out, 1); }
int zk_transcript_get_permutations(zk_transcript_t* t, uint32_t n, uint32_t nbits, uint64_t* out) {
    return guard([&] {
        ZK_REQUIRE(t && out, "transcript: null");
        on_stream(t->stream);
        t->io.reserve((size_t)n * sizeof(u64) + 64);
        transcript_permutations_dev(t->state.p, n, nbits, t->io.u(), t->stream);
        ZK_HIP(hipStreamSynchronize(t->stream));
        ZK_HIP(hipMemcpy(out, t->io.p, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
    });
}
int zk_transcript_free(zk_transcript_t* t) { delete t; return 0; }

// ---- FRI / stark_gen glue ----------------------------------------------------------------------
int zk_fri_fold_dev(const uint64_t* d_pol, uint32_t pol_bits, uint32_t step_bits, const uint64_t* d_special_x,
                    uint64_t shift_inv, uint64_t* d_out, void* stream) {
    return guard([&] { fri_fold_dev((const u64*)d_pol, pol_bits, step_bits, (const u64*)d_special_x, shift_inv, (u64*)d_out, on_stream((hipStream_t)stream)); });
}
int zk_fri_transpose_dev(const uint64_t* d_pol, uint64_t n, uint32_t tbits, uint64_t* d_out, void* stream) {
    return guard([&] { fri_transpose_dev((const u64*)d_pol, n, tbits, (u64*)d_out, on_stream((hipStream_t)stream)); });
}
int zk_stark_x_table_dev(uint32_t nbits, uint64_t shift, uint64_t* d_out, void* stream) {
    return guard([&] { ZK_REQUIRE(nbits <= 32, "x_table: nbits > 32"); x_table_dev(nbits, shift, (u64*)d_out, on_stream((hipStream_t)stream)); });
}
int zk_stark_zh_inv_dev(uint32_t nbits, uint32_t extend_bits, uint64_t* d_out, void* stream) {
    return guard([&] { ZK_REQUIRE(extend_bits <= 16, "zh_inv: extend_bits > 16"); zh_inv_dev(nbits, extend_bits, (u64*)d_out, on_stream((hipStream_t)stream)); });
}
int zk_stark_xdivxsub_dev(const uint64_t* d_xi, uint64_t mulw, uint32_t nbits_ext, uint64_t* d_out, void* stream) {
    return guard([&] { ZK_REQUIRE(nbits_ext <= 32, "xdivxsub: nbits_ext > 32"); xdivxsub_dev((const u64*)d_xi, mulw, nbits_ext, (u64*)d_out, on_stream((hipStream_t)stream)); });
}
int zk_stark_lev_dev(const uint64_t* d_xi, uint32_t nbits, int prime, uint64_t* d_out, uint64_t* d_tmp, uint64_t* d_tmp2, void* stream) {
    return guard([&] { lev_dev((const u64*)d_xi, nbits, prime != 0, 49, (u64*)d_out, (u64*)d_tmp, (u64*)d_tmp2, on_stream((hipStream_t)stream)); });
}
int zk_stark_evals_dev(const zk_eval_desc* descs, uint32_t n_ev, uint32_t nbits, uint32_t ext, const uint64_t* d_LEv,
                       const uint64_t* d_LpEv, uint64_t* d_out, void* stream) {
    return guard([&] {
        static_assert(sizeof(zk_eval_desc) == sizeof(EvalDescHost), "eval descriptor layout");
        evals_dev((const EvalDescHost*)descs, n_ev, nbits, ext, (const u64*)d_LEv, (const u64*)d_LpEv, (u64*)d_out, on_stream((hipStream_t)stream));
    });
}
int zk_stark_qsplit_dev(const uint64_t* d_qq1, uint32_t nbits, uint32_t q_dim, uint32_t q_deg, uint64_t* d_qq2, void* stream) {
    return guard([&] { qsplit_dev((const u64*)d_qq1, nbits, q_dim, q_deg, (u64*)d_qq2, on_stream((hipStream_t)stream)); });
}

int zk_msm_g1_bn254_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream) {
    return guard([&] { msm_g1_bn254_dev(d_bases, d_scalars, n, d_out, on_stream((hipStream_t)stream)); });
}
int zk_g1_bn254_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream) {
    return guard([&] { g1_bn254_mul_generator_dev((const u64*)d_k, n, d_bases, on_stream((hipStream_t)stream)); });
}
int zk_msm_g1_bn254(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity) {
    return guard([&] {
        ZK_REQUIRE(out && is_infinity, "msm: null output");
        ZK_REQUIRE(n == 0 || (bases && scalars), "msm: null input");
        if (n == 0) { memset(out, 0, 64); *is_infinity = 1; return; }  // empty sum
        DevBuf db, ds, dout;
        db.reserve(n * 64); ds.reserve(n * 32); dout.reserve(68);
        ZK_HIP(hipMemcpy(db.p, bases, n * 64, hipMemcpyHostToDevice));
        ZK_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
        msm_g1_bn254_dev(db.p, ds.p, n, dout.p, nullptr);
        uint32_t h[17];
        ZK_HIP(hipStreamSynchronize(nullptr));
        ZK_HIP(hipMemcpy(h, dout.p, 68, hipMemcpyDeviceToHost));
        memcpy(out, h, 64);
        *is_infinity = (int)h[16];
    });
}

// G2 variants: same contract, points of PB bytes
#define ZK_MSM_G2(NAME, PB)                                                                                               \
    int zk_g2_##NAME##_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream) {                 \
        return guard([&] { g2_##NAME##_mul_generator_dev((const u64*)d_k, n, d_bases, on_stream((hipStream_t)stream)); });          \
    }                                                                                                                   \
    int zk_msm_g2_##NAME##_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream) {       \
        return guard([&] { msm_g2_##NAME##_dev(d_bases, d_scalars, n, d_out, on_stream((hipStream_t)stream)); });                   \
    }                                                                                                                   \
    int zk_msm_g2_##NAME(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity) {               \
        return guard([&] {                                                                                              \
            ZK_REQUIRE(out && is_infinity, "msm: null output");                                                        \
            ZK_REQUIRE(n == 0 || (bases && scalars), "msm: null input");                                               \
            if (n == 0) { memset(out, 0, PB); *is_infinity = 1; return; }                                               \
            DevBuf db, ds, dout;                                                                                        \
            db.reserve(n * PB); ds.reserve(n * 32); dout.reserve(PB + 4);                                               \
            ZK_HIP(hipMemcpy(db.p, bases, n * PB, hipMemcpyHostToDevice));                                              \
            ZK_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));                                            \
            msm_g2_##NAME##_dev(db.p, ds.p, n, dout.p, nullptr);                                                        \
            uint32_t h[PB / 4 + 1];                                                                                     \
            ZK_HIP(hipStreamSynchronize(nullptr));                                                                      \
            ZK_HIP(hipMemcpy(h, dout.p, PB + 4, hipMemcpyDeviceToHost));                                                \
            memcpy(out, h, PB);                                                                                         \
            *is_infinity = (int)h[PB / 4];                                                                              \
        });                                                                                                             \
    }
ZK_MSM_G2(bn254, 128)
ZK_MSM_G2(bls12_381, 192)
#undef ZK_MSM_G2

int zk_g1_bls12_381_mul_generator_dev(const uint64_t* d_k, uint64_t n, void* d_bases, void* stream) {
    return guard([&] { g1_bls12_381_mul_generator_dev((const u64*)d_k, n, d_bases, on_stream((hipStream_t)stream)); });
}
int zk_msm_g1_bls12_381_dev(const void* d_bases, const void* d_scalars, uint64_t n, void* d_out, void* stream) {
    return guard([&] { msm_g1_bls12_381_dev(d_bases, d_scalars, n, d_out, on_stream((hipStream_t)stream)); });
}
int zk_msm_g1_bls12_381(const void* bases, const void* scalars, uint64_t n, void* out, int* is_infinity) {
    return guard([&] {
        ZK_REQUIRE(out && is_infinity, "msm: null output");
        ZK_REQUIRE(n == 0 || (bases && scalars), "msm: null input");
        if (n == 0) { memset(out, 0, 96); *is_infinity = 1; return; }  // empty sum
        DevBuf db, ds, dout;
        db.reserve(n * 96); ds.reserve(n * 32); dout.reserve(100);
        ZK_HIP(hipMemcpy(db.p, bases, n * 96, hipMemcpyHostToDevice));
        ZK_HIP(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
        msm_g1_bls12_381_dev(db.p, ds.p, n, dout.p, nullptr);
        uint32_t h[25];
        ZK_HIP(hipStreamSynchronize(nullptr));
        ZK_HIP(hipMemcpy(h, dout.p, 100, hipMemcpyDeviceToHost));
        memcpy(out, h, 96);
        *is_infinity = (int)h[24];
    });
}

// ---- scalar-field hashing (verificationHashType "BN128" / "BLS12381") -----------------------------------
}  // extern "C" (reopened below): the two fields share one implementation, instantiated per field

namespace {
struct FrOps {   // one scalar field: device entry points (frhash.hip) + the host-side modulus for the sponge bookkeeping
    const char* name;
    uint64_t R[4], R2[4], INV;     // modulus, 2^512 mod r, -r^-1 mod 2^64
    void (*load)(const char*);
    std::string (*selfcheck)(const char*);
    void (*poseidon_dev)(const u64*, uint64_t, uint32_t, const u64*, uint32_t, u64*, hipStream_t);
    uint64_t (*n_nodes)(uint64_t);
    void (*linearhash_rows_dev)(const u64*, uint32_t, uint64_t, u64*, hipStream_t);
    void (*merkelize_dev)(const u64*, uint32_t, uint64_t, u64*, hipStream_t);
};
const FrOps FR_BN128 = {"bn128",
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {1997599621687373223ULL, 6052339484930628067ULL, 10108755138030829701ULL, 150537098327114917ULL}, 0xc2e1f593efffffffULL,
    bn128_load_constants, bn128_tables_selfcheck, bn128_poseidon_dev, bn128_merkle_n_nodes, bn128_linearhash_rows_dev, bn128_merkelize_dev};
const FrOps FR_BLS12381 = {"bls12381",
    {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
    {14526898881837571181ULL, 3129137299524312099ULL, 419701826671360399ULL, 524908885293268753ULL}, 0xfffffffeffffffffULL,
    bls12381_load_constants, bls12381_tables_selfcheck, bls12381_poseidon_dev, bls12381_merkle_n_nodes, bls12381_linearhash_rows_dev, bls12381_merkelize_dev};

struct FrMerkle {
    const FrOps* F = nullptr;
    const u64* d_elements = nullptr;
    DevBuf owned_elements, nodes;
    uint32_t width = 0, depth = 0;
    uint64_t height = 0, n_nodes = 0;
};
struct FrTranscript {   // transcript_bn128.rs:14-20: sponge bookkeeping on the host, permutations on the device
    const FrOps* F = nullptr;
    uint64_t state[4] = {0, 0, 0, 0};
    std::vector<uint64_t> pending;          // raw limbs, 4 per element
    std::vector<uint64_t> out;              // 17 x 4 raw limbs
    size_t out_pos = 0, n_out = 0;
    uint64_t out3[3] = {0, 0, 0}; size_t out3_pos = 0, n_out3 = 0;
    DevBuf d_in, d_init, d_out;
};
uint32_t fr_depth(uint64_t height) { uint32_t d = 0; uint64_t n = height; while (n > 1) { n = (n - 1) / 16 + 1; ++d; } return d; }
void fr_mont_mul_host(const FrOps& F, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]) {   // a*b/2^256 mod r (CIOS)
    typedef unsigned __int128 u128;
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * F.INV;
        c = ((u128)m * F.R[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * F.R[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    for (;;) {
        bool ge = t[4] != 0;
        if (!ge) { ge = true; for (int i = 3; i >= 0; --i) { if (t[i] > F.R[i]) break; if (t[i] < F.R[i]) { ge = false; break; } } }
        if (!ge) break;
        u128 br = 0;
        for (int i = 0; i < 4; ++i) { u128 d = (u128)t[i] - F.R[i] - br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
        t[4] -= (uint64_t)br;
    }
    memcpy(r, t, 32);
}
}  // namespace
namespace zk {
// decimal text of a scalar-field element (how a digest travels in zkin JSON, digest.rs:91-94) -> the raw Montgomery limbs of
// ElementDigest<4, Fr>; false when the text is not a canonical value below the modulus
bool fr_digest_from_dec(bool bls12381, const std::string& dec, u64 out[4]) {
    const FrOps& F = bls12381 ? FR_BLS12381 : FR_BN128;
    if (dec.empty() || dec.size() > 78) return false;
    uint64_t v[4] = {0, 0, 0, 0};
    for (char c : dec) {
        if (c < '0' || c > '9') return false;
        unsigned __int128 carry = (unsigned)(c - '0');
        for (int i = 0; i < 4; ++i) { carry += (unsigned __int128)v[i] * 10; v[i] = (uint64_t)carry; carry >>= 64; }
        if (carry) return false;
    }
    for (int i = 3; i >= 0; --i) { if (v[i] < F.R[i]) break; if (v[i] > F.R[i] || i == 0) return false; }
    uint64_t m[4];
    fr_mont_mul_host(F, v, F.R2, m);
    for (int i = 0; i < 4; ++i) out[i] = m[i];
    return true;
}
// n hashes of 16 nodes each (hash of a group of a 16-ary tree with a zero initial state, merklehash_bn128.rs:108-128), device buffers
// d_out [n][2][4]: the permutation's first two words -- Poseidon::hash is word 0 over BN128 (poseidon_bn128_opt.rs) and word 1 over
// BLS12-381 (poseidon_bls12381_opt.rs:94-103)
void fr_hash16_dev(bool bls12381, const u64* d_in /* [n][16][4] */, uint64_t n, const u64* d_zero4, u64* d_out, hipStream_t st) {
    (bls12381 ? FR_BLS12381 : FR_BN128).poseidon_dev(d_in, n, 16, d_zero4, 2, d_out, st);
}
void fr_linearhash_rows_dev(bool bls12381, const u64* d_rows, uint32_t width, uint64_t height, u64* d_digests, hipStream_t st) {
    (bls12381 ? FR_BLS12381 : FR_BN128).linearhash_rows_dev(d_rows, width, height, d_digests, st);
}
}  // namespace zk
namespace {
void fr_tr_update(FrTranscript* t) {   // transcript_bn128.rs:22-31
    t->pending.resize(64, 0);
    on_stream(nullptr);                         // the scalar-field sponges work on the null stream, whoever calls
    t->d_in.reserve(64 * 8); t->d_init.reserve(32); t->d_out.reserve(17 * 32);
    ZK_HIP(hipMemcpy(t->d_in.p, t->pending.data(), 64 * 8, hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(t->d_init.p, t->state, 32, hipMemcpyHostToDevice));
    t->F->poseidon_dev(t->d_in.u(), 1, 16, t->d_init.u(), 17, t->d_out.u(), nullptr);
    t->out.resize(68);
    ZK_HIP(hipStreamSynchronize(nullptr));
    ZK_HIP(hipMemcpy(t->out.data(), t->d_out.p, 17 * 32, hipMemcpyDeviceToHost));
    t->out_pos = 0; t->n_out = 17; t->n_out3 = 0; t->out3_pos = 0; t->pending.clear();
    memcpy(t->state, t->out.data(), 32);
}
void fr_tr_add1(FrTranscript* t, const uint64_t raw[4]) {   // :32-40
    t->n_out = 0; t->out_pos = 0;
    t->pending.insert(t->pending.end(), raw, raw + 4);
    if (t->pending.size() == 64) fr_tr_update(t);
}
void fr_tr_get253(FrTranscript* t, uint64_t canon[4]) {   // :42-48, canonical value of the popped element
    if (t->out_pos >= t->n_out) fr_tr_update(t);
    const uint64_t one[4] = {1, 0, 0, 0};
    fr_mont_mul_host(*t->F, t->out.data() + 4 * t->out_pos, one, canon);
    ++t->out_pos;
}

int fr_poseidon(const FrOps& F, const uint64_t* inp, uint32_t n_in, const uint64_t* init_state, uint32_t n_out, uint64_t* out) {
    return guard([&] {
        ZK_REQUIRE(inp && init_state && out, "poseidon: null buffer");
        ZK_REQUIRE(n_in >= 1 && n_in <= 16, "Wrong inputs length");
        on_stream(nullptr);
        DevBuf d_in, d_init, d_out;
        d_in.reserve(n_in * 32); d_init.reserve(32); d_out.reserve(17 * 32);
        ZK_HIP(hipMemcpy(d_in.p, inp, n_in * 32, hipMemcpyHostToDevice));
        ZK_HIP(hipMemcpy(d_init.p, init_state, 32, hipMemcpyHostToDevice));
        F.poseidon_dev(d_in.u(), 1, n_in, d_init.u(), n_out, d_out.u(), nullptr);
        ZK_HIP(hipStreamSynchronize(nullptr));
        ZK_HIP(hipMemcpy(out, d_out.p, n_out * 32, hipMemcpyDeviceToHost));
    });
}
int fr_linearhash(const FrOps& F, const uint64_t* v, size_t n, uint64_t* out) {
    return guard([&] {
        ZK_REQUIRE(out && (v || n == 0), "linearhash: null buffer");
        on_stream(nullptr);
        DevBuf d_v, d_o; d_v.reserve(n * 8 + 8); d_o.reserve(32);
        if (n) ZK_HIP(hipMemcpy(d_v.p, v, n * 8, hipMemcpyHostToDevice));
        ZK_HIP(hipMemset(d_o.p, 0, 32));
        F.linearhash_rows_dev(d_v.u(), (uint32_t)n, 1, d_o.u(), nullptr);
        ZK_HIP(hipStreamSynchronize(nullptr));
        ZK_HIP(hipMemcpy(out, d_o.p, 32, hipMemcpyDeviceToHost));
    });
}
template <class T>
T* fr_merkelize(const FrOps& F, const uint64_t* buff, bool on_device, uint32_t width, uint64_t height, void* stream) {
    T* t = nullptr;
    if (guard([&] {
            ZK_REQUIRE(height >= 1, "merkelize: height must be >= 1");
            ZK_REQUIRE(buff || width == 0, "merkelize: null buffer");
            t = new T;
            t->F = &F;
            if (on_device) t->d_elements = (const u64*)buff;
            else {
                on_stream(nullptr);
                t->owned_elements.reserve((size_t)width * height * 8 + 8);
                if (width) ZK_HIP(hipMemcpy(t->owned_elements.p, buff, (size_t)width * height * 8, hipMemcpyHostToDevice));
                t->d_elements = t->owned_elements.u();
            }
            t->width = width; t->height = height;
            t->n_nodes = F.n_nodes(height); t->depth = fr_depth(height);
            t->nodes.reserve(t->n_nodes * 32);
            F.merkelize_dev(t->d_elements, width, height, t->nodes.u(), on_device ? on_stream((hipStream_t)stream) : nullptr);
            if (!on_device) ZK_HIP(hipStreamSynchronize(nullptr));
        }) != 0) { delete t; return nullptr; }
    return t;
}
int fr_merkle_root(const FrMerkle* t, uint64_t* out) {
    return guard([&] {
        ZK_REQUIRE(t && out, "null argument");
        ZK_HIP(hipDeviceSynchronize());
        ZK_HIP(hipMemcpy(out, t->nodes.u() + 4 * (t->n_nodes - 1), 32, hipMemcpyDeviceToHost));
    });
}
int fr_merkle_nodes(const FrMerkle* t, uint64_t* out) {
    return guard([&] { ZK_REQUIRE(t && out, "null argument"); ZK_HIP(hipDeviceSynchronize()); ZK_HIP(hipMemcpy(out, t->nodes.p, t->n_nodes * 32, hipMemcpyDeviceToHost)); });
}
// get_group_proof (merklehash_bn128.rs:246-254): row_out[width], path_out[depth][16][4]
int fr_merkle_group_proof(const FrMerkle* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out) {
    return guard([&] {
        ZK_REQUIRE(t && row_out && path_out, "null argument");
        ZK_REQUIRE(idx < t->height, "MerkleTreeError: access invalid node");
        ZK_HIP(hipDeviceSynchronize());
        if (t->width) ZK_HIP(hipMemcpy(row_out, t->d_elements + idx * t->width, t->width * 8, hipMemcpyDeviceToHost));
        uint64_t n = t->height, off = 0, id = idx; uint32_t d = 0;
        while (n > 1) {   // merklehash_bn128.rs:86-106
            const uint64_t si = id & ~(uint64_t)15;
            ZK_HIP(hipMemcpy(path_out + (size_t)d * 64, t->nodes.u() + 4 * (off + si), 16 * 32, hipMemcpyDeviceToHost));
            const uint64_t next = (n - 1) / 16 + 1;
            off += next * 16; n = next; id >>= 4; ++d;
        }
    });
}
int fr_merkle_group_proofs(const FrMerkle* t, const uint64_t* idx, uint32_t n, uint64_t* rows_out, uint64_t* paths_out) {
    return guard([&] {
        ZK_REQUIRE(t && (n == 0 || (idx && rows_out && paths_out)), "null argument");
        if (n == 0) return;
        for (uint32_t q = 0; q < n; ++q) ZK_REQUIRE(idx[q] < t->height, "MerkleTreeError: access invalid node");
        ZK_HIP(hipDeviceSynchronize());                                   // (the tree does not remember the stream it was built on)
        on_stream(nullptr);
        const size_t per = (size_t)t->width + 64 * (size_t)t->depth;
        DevBuf d_idx, d_out; d_idx.reserve(n * 8); d_out.reserve(std::max<size_t>(1, per * n) * 8);
        ZK_HIP(hipMemcpy(d_idx.p, idx, n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(fr_gather_proofs_kernel, dim3(n), dim3(64), 0, nullptr, t->d_elements, t->nodes.u(), t->width, t->height,
                           t->depth, d_idx.u(), d_out.u());
        ZK_HIP(hipGetLastError());
        std::vector<u64> h(std::max<size_t>(1, per * n));
        ZK_HIP(hipMemcpy(h.data(), d_out.p, per * n * 8, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < n; ++q) {
            memcpy(rows_out + (size_t)q * t->width, h.data() + q * per, (size_t)t->width * 8);
            if (t->depth) memcpy(paths_out + (size_t)q * 64 * t->depth, h.data() + q * per + t->width, (size_t)t->depth * 512);
        }
    });
}
// put (transcript_bn128.rs:90-101): n == 1 -> a Goldilocks value, n == 4 -> a digest (raw limbs); anything else is an error
int fr_transcript_put(FrTranscript* t, const uint64_t* e, size_t n) {
    return guard([&] {
        ZK_REQUIRE(t && e, "null argument");
        if (n == 1) {
            const uint64_t x[4] = {e[0], 0, 0, 0}; uint64_t m[4];
            fr_mont_mul_host(*t->F, x, t->F->R2, m);
            fr_tr_add1(t, m);
        } else if (n == 4) fr_tr_add1(t, e);
        else throw Error("Invalid elements as inputs to transcript");
    });
}
int fr_transcript_get_fields1(FrTranscript* t, uint64_t* out) {   // :71-88
    return guard([&] {
        ZK_REQUIRE(t && out, "null argument");
        for (;;) {
            if (t->out3_pos < t->n_out3) { *out = t->out3[t->out3_pos++]; return; }
            if (t->out_pos < t->n_out) {
                uint64_t c[4];
                fr_tr_get253(t, c);
                for (int i = 0; i < 3; ++i) t->out3[i] = c[i] % GL_P;   // biguint_to_be (helper.rs:61-65)
                t->out3_pos = 0; t->n_out3 = 3;
                continue;
            }
            fr_tr_update(t);
        }
    });
}
int fr_transcript_get_permutations(FrTranscript* t, uint32_t n, uint32_t nbits, uint64_t* out) {   // :103-131
    return guard([&] {
        ZK_REQUIRE(t && out && n >= 1 && nbits >= 1 && nbits <= 63, "bad argument");
        const uint32_t total = n * nbits, nf = (total - 1) / 253 + 1;
        std::vector<uint64_t> f(4 * (size_t)nf);
        for (uint32_t i = 0; i < nf; ++i) fr_tr_get253(t, f.data() + 4 * i);
        uint32_t cf = 0, cb = 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t a = 0;
            for (uint32_t j = 0; j < nbits; ++j) {
                if ((f[4 * cf + cb / 64] >> (cb % 64)) & 1) a += (uint64_t)1 << j;
                if (++cb == 253) { cb = 0; ++cf; }
            }
            out[i] = a;
        }
    });
}
}  // namespace

struct zk_bn128_merkle : FrMerkle {};
struct zk_bn128_transcript : FrTranscript {};
struct zk_bls12381_merkle : FrMerkle {};
struct zk_bls12381_transcript : FrTranscript {};

extern "C" {
#define ZK_FRHASH_CAPI(P, OPS)                                                                                          \
    int zk_##P##_load_constants(const char* path) { return guard([&] { ZK_REQUIRE(path, "null path"); OPS.load(path); }); } \
    int zk_##P##_poseidon_selfcheck(const char* path) {                  /* host arithmetic only: no CallScope, no device */ \
        try {                                                                                                           \
            if (!path) { set_error("null path"); return -1; }                                                           \
            const std::string why = OPS.selfcheck(path);                                                                \
            if (why.empty()) return 0;                                                                                  \
            set_error(why);                                                                                             \
        } catch (const std::exception& e) { set_error(e.what()); }                                                      \
        return -1;                                                                                                      \
    }                                                                                                                   \
    int zk_##P##_poseidon(const uint64_t* inp, uint32_t n_in, const uint64_t init_state[4], uint32_t n_out, uint64_t* out) { \
        return fr_poseidon(OPS, inp, n_in, init_state, n_out, out);                                                     \
    }                                                                                                                   \
    int zk_##P##_poseidon_dev(const uint64_t* d_inp, uint64_t n, uint32_t n_in, const uint64_t* d_init_state, uint32_t n_out, \
                              uint64_t* d_out, void* stream) {                                                          \
        return guard([&] { OPS.poseidon_dev((const u64*)d_inp, n, n_in, (const u64*)d_init_state, n_out, (u64*)d_out, on_stream((hipStream_t)stream)); }); \
    }                                                                                                                   \
    int zk_##P##_linearhash(const uint64_t* v, size_t n, uint64_t out[4]) { return fr_linearhash(OPS, v, n, out); }       \
    uint64_t zk_##P##_merkle_n_nodes(uint64_t height) { return height ? OPS.n_nodes(height) : 0; }                        \
    zk_##P##_merkle_t* zk_##P##_merkelize(const uint64_t* buff, uint32_t width, uint64_t height) {                        \
        return fr_merkelize<zk_##P##_merkle>(OPS, buff, false, width, height, nullptr);                                  \
    }                                                                                                                   \
    zk_##P##_merkle_t* zk_##P##_merkelize_dev(const uint64_t* d_buff, uint32_t width, uint64_t height, void* stream) {    \
        return fr_merkelize<zk_##P##_merkle>(OPS, d_buff, true, width, height, stream);                                  \
    }                                                                                                                   \
    int zk_##P##_merkle_root(const zk_##P##_merkle_t* t, uint64_t out[4]) { return fr_merkle_root(t, out); }              \
    int zk_##P##_merkle_nodes(const zk_##P##_merkle_t* t, uint64_t* out) { return fr_merkle_nodes(t, out); }              \
    uint32_t zk_##P##_merkle_depth(const zk_##P##_merkle_t* t) { return t ? t->depth : 0; }                               \
    int zk_##P##_merkle_group_proof(const zk_##P##_merkle_t* t, uint64_t idx, uint64_t* row_out, uint64_t* path_out) {    \
        return fr_merkle_group_proof(t, idx, row_out, path_out);                                                         \
    }                                                                                                                    \
    int zk_##P##_merkle_group_proofs(const zk_##P##_merkle_t* t, const uint64_t* idx, uint32_t n, uint64_t* rows_out,     \
                                     uint64_t* paths_out) {                                                              \
        return fr_merkle_group_proofs(t, idx, n, rows_out, paths_out);                                                   \
    }                                                                                                                   \
    int zk_##P##_merkle_free(zk_##P##_merkle_t* t) { return guard([&] { delete t; }); }                                   \
    zk_##P##_transcript_t* zk_##P##_transcript_new(void) {                                                               \
        zk_##P##_transcript* t = nullptr;                                                                               \
        if (guard([&] { t = new zk_##P##_transcript; t->F = &OPS; }) != 0) return nullptr;                               \
        return t;                                                                                                       \
    }                                                                                                                   \
    int zk_##P##_transcript_put(zk_##P##_transcript_t* t, const uint64_t* e, size_t n) { return fr_transcript_put(t, e, n); } \
    int zk_##P##_transcript_get_fields1(zk_##P##_transcript_t* t, uint64_t* out) { return fr_transcript_get_fields1(t, out); } \
    int zk_##P##_transcript_get_field(zk_##P##_transcript_t* t, uint64_t out[3]) {                                        \
        for (int i = 0; i < 3; ++i) { const int rc = fr_transcript_get_fields1(t, out + i); if (rc) return rc; }         \
        return 0;                                                                                                       \
    }                                                                                                                   \
    int zk_##P##_transcript_get_permutations(zk_##P##_transcript_t* t, uint32_t n, uint32_t nbits, uint64_t* out) {       \
        return fr_transcript_get_permutations(t, n, nbits, out);                                                         \
    }                                                                                                                   \
    int zk_##P##_transcript_free(zk_##P##_transcript_t* t) { return guard([&] { delete t; }); }
ZK_FRHASH_CAPI(bn128, FR_BN128)
ZK_FRHASH_CAPI(bls12381, FR_BLS12381)
#undef ZK_FRHASH_CAPI

int zk_stark_get_pol_dev(const uint64_t* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n, uint64_t* d_out3, void* stream) {
    return guard([&] { pol_get_dev((const u64*)d_buf, width, offset, dim, n, (u64*)d_out3, on_stream((hipStream_t)stream)); });
}
int zk_stark_set_pol_dev(uint64_t* d_buf, uint64_t width, uint64_t offset, uint32_t dim, uint64_t n, const uint64_t* d_in3, void* stream) {
    return guard([&] { pol_set_dev((u64*)d_buf, width, offset, dim, n, (const u64*)d_in3, on_stream((hipStream_t)stream)); });
}
int zk_stark_calculate_h1h2_dev(const uint64_t* d_f3, const uint64_t* d_t3, uint64_t n, uint64_t* d_h1_3, uint64_t* d_h2_3, void* stream) {
    return guard([&] {
        ZK_REQUIRE(d_f3 && d_t3 && d_h1_3 && d_h2_3 && n >= 1, "calculate_H1H2: null or empty argument");
        ZK_REQUIRE(n <= (1ull << 28), "calculate_H1H2: more than 2^28 rows");
        hipStream_t st = on_stream((hipStream_t)stream);
        DevBuf work; work.reserve(h1h2_work_words(n) * 8);
        u64* d_missing = nullptr;
        calculate_h1h2_dev((const u64*)d_f3, (const u64*)d_t3, n, (u64*)d_h1_3, (u64*)d_h2_3, work.u(), &d_missing, st);
        u64 missing = 0;
        ZK_HIP(hipStreamSynchronize(st));
        ZK_HIP(hipMemcpy(&missing, d_missing, 8, hipMemcpyDeviceToHost));
        if (missing != ~0ull) {                                          // stark_gen.rs:636-638, the first f that the table lacks
            u64 e = 0;
            ZK_HIP(hipMemcpy(&e, d_f3 + 3 * missing, 8, hipMemcpyDeviceToHost));
            throw Error("Number not included: " + std::to_string(e));
        }
    });
}
int zk_stark_calculate_z_dev(const uint64_t* d_num3, const uint64_t* d_den3, uint64_t n, uint64_t* d_z3, void* stream) {
    return guard([&] {
        ZK_REQUIRE(n >= 1, "calculate_Z: empty polynomial");
        DevBuf work; work.reserve((n + n / 1024 + 8) * 24);
        u64 h[3];
        calculate_z_dev((const u64*)d_num3, (const u64*)d_den3, n, (u64*)d_z3, work.u(), work.u() + 3 * (n + n / 1024 + 4), on_stream((hipStream_t)stream));
        ZK_HIP(hipStreamSynchronize(on_stream((hipStream_t)stream)));
        ZK_HIP(hipMemcpy(h, work.u() + 3 * (n + n / 1024 + 4), 24, hipMemcpyDeviceToHost));
        ZK_REQUIRE(h[0] == 1 && h[1] == 0 && h[2] == 0, "calculate_Z: z does not close (grand product != 1)");
    });
}

const uint64_t* zk_merkle_elements_dev(const zk_merkle_t* t) { return t ? (const uint64_t*)t->d_elements : nullptr; }
const uint64_t* zk_merkle_nodes_dev(const zk_merkle_t* t) { return t ? (const uint64_t*)t->nodes.p : nullptr; }
int zk_merkle_free(zk_merkle_t* t) { delete t; return 0; }

#define ZK_MSM_TABLE_API(NAME)                                                                                              \
    size_t zk_msm_##NAME##_table_bytes(uint64_t table_n) { return msm_##NAME##_fixed_table_bytes(table_n); }                \
    int zk_msm_##NAME##_table_build_dev(const void* d_bases, uint64_t table_n, void* d_table, void* stream) {               \
        return guard([&] { ZK_REQUIRE(d_bases && d_table, "msm table: null argument"); msm_##NAME##_fixed_prepare_dev(d_bases, table_n, d_table, on_stream((hipStream_t)stream)); }); \
    }                                                                                                                       \
    int zk_msm_##NAME##_table_dev(const void* d_table, uint64_t table_n, uint64_t offset, const void* d_scalars, uint64_t n, void* d_out, void* stream) { \
        return guard([&] { ZK_REQUIRE(d_table && d_scalars && d_out, "msm table: null argument"); msm_##NAME##_fixed_dev(d_table, table_n, offset, d_scalars, n, d_out, on_stream((hipStream_t)stream)); }); \
    }
ZK_MSM_TABLE_API(g1_bn254)
ZK_MSM_TABLE_API(g2_bn254)
ZK_MSM_TABLE_API(g1_bls12_381)
ZK_MSM_TABLE_API(g2_bls12_381)
#undef ZK_MSM_TABLE_API
// ---- compressor12 exec (compressor12.hip) ----------------------------------------------------------------------
struct zk_c12_exec { C12Exec* impl; };
zk_c12_exec_t* zk_c12_exec_new(const char* exec_json, size_t len, uint64_t n_witness) {
    zk_c12_exec_t* out = nullptr;
    if (guard([&] { C12Exec* e = c12_exec_new(exec_json, len, n_witness); out = new zk_c12_exec{e}; }) != 0) return nullptr;
    return out;
}
int zk_c12_exec_dev(const zk_c12_exec_t* e, const uint64_t* d_witness, uint64_t n_witness, uint64_t n_rows, uint64_t* d_cm, void* stream) {
    return guard([&] { ZK_REQUIRE(e && e->impl, "compressor12: null handle"); c12_exec_dev(e->impl, (const u64*)d_witness, n_witness, n_rows, (u64*)d_cm, on_stream((hipStream_t)stream)); });
}
uint64_t zk_c12_exec_depth(const zk_c12_exec_t* e) { return e && e->impl ? c12_exec_levels(e->impl) : 0; }
int zk_c12_exec_free(zk_c12_exec_t* e) {
    return guard([&] { if (e) { c12_exec_free(e->impl); delete e; } });
}

// ---- Groth16 (groth16.hip) ------------------------------------------------------------------------------------
#define ZK_FR_NTT(NAME)                                                                                                  \
    int zk_fr_##NAME##_ntt_dev(uint64_t* d, uint32_t log_n, int inverse, int coset, void* stream) {                      \
        return guard([&] { ZK_REQUIRE(d, "fr ntt: null data"); fr_##NAME##_ntt_dev((u64*)d, (int)log_n, inverse != 0, coset != 0, on_stream((hipStream_t)stream)); }); \
    }                                                                                                                    \
    int zk_fr_##NAME##_ntt(uint64_t* data, uint32_t log_n, int inverse, int coset) {                                     \
        return guard([&] {                                                                                               \
            ZK_REQUIRE(data, "fr ntt: null data");                                                                       \
            ZK_REQUIRE(log_n <= 32, "fr ntt: domain too large");                                                         \
            const size_t bytes = ((size_t)32) << log_n;                                                                  \
            DevBuf d; d.reserve(bytes);                                                                                  \
            ZK_HIP(hipMemcpy(d.p, data, bytes, hipMemcpyHostToDevice));                                                  \
            fr_##NAME##_ntt_dev((u64*)d.p, (int)log_n, inverse != 0, coset != 0, nullptr);                               \
            ZK_HIP(hipStreamSynchronize(nullptr));                                                                       \
            ZK_HIP(hipMemcpy(data, d.p, bytes, hipMemcpyDeviceToHost));                                                  \
        });                                                                                                              \
    }                                                                                                                    \
    int zk_fr_##NAME##_quotient_dev(uint64_t* a, const uint64_t* b, const uint64_t* c, uint32_t log_n, void* stream) {   \
        return guard([&] { ZK_REQUIRE(a && b && c, "fr quotient: null data"); fr_##NAME##_quotient_dev((u64*)a, (const u64*)b, (const u64*)c, (int)log_n, on_stream((hipStream_t)stream)); }); \
    }
ZK_FR_NTT(bn254)
ZK_FR_NTT(bls12_381)
#undef ZK_FR_NTT

int zk_fq_bn254_convert_dev(void* d, uint64_t n, int to_mont, void* stream) {
    return guard([&] { ZK_REQUIRE(d || n == 0, "fq convert: null data"); if (to_mont) fq_bn254_canon_to_mont_dev(d, n, on_stream((hipStream_t)stream)); else fq_bn254_mont_to_canon_dev(d, n, on_stream((hipStream_t)stream)); });
}
int zk_fq_bls12_381_convert_dev(void* d, uint64_t n, int to_mont, void* stream) {
    return guard([&] { ZK_REQUIRE(d || n == 0, "fq convert: null data"); if (to_mont) fq_bls12_381_canon_to_mont_dev(d, n, on_stream((hipStream_t)stream)); else fq_bls12_381_mont_to_canon_dev(d, n, on_stream((hipStream_t)stream)); });
}
struct zk_groth16_setup { Groth16Setup* impl; };
zk_groth16_setup_t* zk_groth16_setup_new(const char* curve, const void* r1cs, size_t r1cs_len, const void* params, size_t params_len) {
    zk_groth16_setup_t* out = nullptr;
    if (guard([&] { Groth16Setup* g = groth16_setup_new(curve, r1cs, r1cs_len, params, params_len); out = new zk_groth16_setup{g}; }) != 0) return nullptr;
    return out;
}
int zk_groth16_setup_info(const zk_groth16_setup_t* s, uint32_t* n_wires, uint32_t* n_inputs, uint32_t* domain_log) {
    return guard([&] {
        ZK_REQUIRE(s && s->impl, "groth16: null setup");
        if (n_wires) *n_wires = s->impl->num_wires();
        if (n_inputs) *n_inputs = s->impl->num_inputs();
        if (domain_log) *domain_log = s->impl->domain_log();
    });
}
static char* groth16_prove_any(zk_groth16_setup_t* s, const void* witness, bool on_device, uint64_t n_wires, const uint64_t* r, const uint64_t* s_, void* proof, uint64_t* d_h) {
    char* out = nullptr;
    if (guard([&] {
            ZK_REQUIRE(s && s->impl, "groth16: null setup");
            ZK_REQUIRE(witness && r && s_, "groth16: null argument");
            Groth16Setup* g = s->impl;
            ZK_REQUIRE(n_wires == g->num_wires(), "groth16: the witness has " + std::to_string(n_wires) + " values, the circuit has " + std::to_string(g->num_wires()) + " wires");
            auto lt = [&](const u32* v) { for (int i = 7; i >= 0; --i) { if (v[i] < g->modulus[i]) return true; if (v[i] > g->modulus[i]) return false; } return false; };
            ZK_REQUIRE(lt((const u32*)r) && lt((const u32*)s_), "groth16: r and s must be canonical field elements");
            if (!on_device) {   // Fr::from_repr (reader.rs:131-134) rejects non-canonical values
                const u32* w = (const u32*)witness;
                for (uint64_t i = 0; i < n_wires; ++i) ZK_REQUIRE(lt(w + 8 * i), "groth16: witness value " + std::to_string(i) + " is not a canonical field element");
            }
            std::string js;
            g->prove(witness, on_device, (const u64*)r, (const u64*)s_, (u32*)proof, &js, (u64*)d_h);
            out = (char*)malloc(js.size() + 1);
            ZK_REQUIRE(out, "out of memory");
            memcpy(out, js.c_str(), js.size() + 1);
        }) != 0) return nullptr;
    return out;
}
char* zk_groth16_prove(zk_groth16_setup_t* s, const void* witness, uint64_t n_wires, const uint64_t r[4], const uint64_t s_[4], void* proof) {
    return groth16_prove_any(s, witness, false, n_wires, r, s_, proof, nullptr);
}
char* zk_groth16_prove_dev(zk_groth16_setup_t* s, const void* d_witness, uint64_t n_wires, const uint64_t r[4], const uint64_t s_[4], void* proof, uint64_t* d_h) {
    return groth16_prove_any(s, d_witness, true, n_wires, r, s_, proof, d_h);
}
int zk_groth16_wtns_payload(const void* wtns, size_t len, const char* curve, uint64_t* offset, uint64_t* n_values) {
    return guard([&] { ZK_REQUIRE(wtns && offset && n_values, "wtns: null argument"); groth16_wtns_payload(wtns, len, curve, offset, n_values); });
}
int zk_groth16_setup_free(zk_groth16_setup_t* s) {
    return guard([&] { if (s) { delete s->impl; delete s; } });
}

}  // extern "C"
