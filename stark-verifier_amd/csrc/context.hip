// Context, stream, scratch allocator, host/device buffer staging and cached tables of libgl355.
#include "gl355_internal.h"
#include <chrono>
#include <time.h>
#include <cstdlib>

namespace gl355 {

int32_t Ctx::fail(int32_t code, const char* msg) {
    err = msg ? msg : "";
    return code;
}
int32_t Ctx::fail_hip(hipError_t e, const char* expr, const char* file, int line) {
    err = std::string("HIP error '") + hipGetErrorString(e) + "' in " + expr + " at " + file + ":" + std::to_string(line);
    if (e == hipErrorOutOfMemory) return GL355_E_OOM;
    return GL355_E_HIP;
}

uint64_t thread_cpu_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
hipError_t Ctx::wait() {
    if (prof_on) {      // host-side accounting: wall time this context's thread spends waiting for its stream, and the CPU time of that wait
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t c0 = thread_cpu_ns();
        const hipError_t e = wait_impl();
        wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        wait_cpu_ns += thread_cpu_ns() - c0;
        wait_calls++;
        return e;
    }
    return wait_impl();
}
hipError_t Ctx::d2h(void* dst, const void* src, size_t bytes) {
    if (!prof_on) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
}
// Three ways to wait for the stream:
//   0  hipStreamSynchronize: the runtime's wait (a spin unless the device runs hipDeviceScheduleBlockingSync) -- lowest latency,
//      one core per waiting context;
//   1  a blocking event (the runtime sleeps on the completion interrupt);
//   2  poll + back-off: hipStreamQuery in a loop, spinning for the first ~20 us and then sleeping 30 us between polls -- a waiting
//      context costs a few percent of a core and wakes within ~30 us of completion, whatever the runtime's interrupt path does.
//      This is what lets a rank run more prover contexts than it has cores.
//   3  poll without ever sleeping: the latency setting of a lone proof (every wait ends within a poll of the completion; a sleep of
//      30 us is 80 us and more with the default timer slack, and a unit has ~40 waits)
hipError_t Ctx::wait_impl() {
    if (blocking_sync == 2 || blocking_sync == 3) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            (void)hipGetLastError();
            if (blocking_sync == 2 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(20)) {
                struct timespec ts = {0, 30000};
                nanosleep(&ts, nullptr);
            }
        }
    }
    if (!blocking_sync) return hipStreamSynchronize(stream);
    if (!sync_ev) {
        hipError_t e = hipEventCreateWithFlags(&sync_ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventRecord(sync_ev, stream);
    if (e != hipSuccess) return e;
    return hipEventSynchronize(sync_ev);
}

int32_t Ctx::aux_stream_get(hipStream_t* out) {
    if (!aux_stream) GL355_HIP(this, hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    *out = aux_stream;
    return GL355_OK;
}
int32_t Ctx::order_event(size_t i, hipEvent_t* out) {
    while (order_ev.size() <= i) {
        hipEvent_t e = nullptr;
        GL355_HIP(this, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        order_ev.push_back(e);
    }
    *out = order_ev[i];
    return GL355_OK;
}

hipEvent_t Ctx::prof_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}

int32_t Ctx::alloc(size_t bytes, void** out) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~size_t(255);
    // best fit among free cached blocks (at most 2x waste)
    int best = -1;
    for (size_t i = 0; i < blocks.size(); i++) {
        if (!blocks[i].used && blocks[i].size >= bytes && blocks[i].size <= 2 * bytes + (1 << 20)) {
            if (best < 0 || blocks[i].size < blocks[best].size) best = (int)i;
        }
    }
    if (best >= 0) { blocks[best].used = true; *out = blocks[best].p; return GL355_OK; }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop the cache and retry once
        (void)hipGetLastError();
        for (auto it = blocks.begin(); it != blocks.end();) {
            if (!it->used) { (void)hipFree(it->p); it = blocks.erase(it); } else ++it;
        }
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail_hip(e, "hipMalloc(scratch)", __FILE__, __LINE__); }
    }
    blocks.push_back({p, bytes, true, block_serial++});
    *out = p;
    return GL355_OK;
}
void Ctx::release(void* p) {
    for (auto& b : blocks)
        if (b.p == p) { b.used = false; return; }
}
int32_t Ctx::pinned(size_t bytes, void** out) {
    if (bytes > pinned_size) {
        if (pinned_buf) { (void)wait_impl(); (void)hipHostFree(pinned_buf); pinned_buf = nullptr; pinned_size = 0; }
        const size_t want = (bytes + (1 << 20)) & ~size_t((1 << 20) - 1);
        hipError_t e = hipHostMalloc(&pinned_buf, want, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); pinned_buf = nullptr; return fail_hip(e, "hipHostMalloc(staging)", __FILE__, __LINE__); }
        pinned_size = want;
    }
    *out = pinned_buf;
    return GL355_OK;
}
void Ctx::runtime_buffers_free() {
    for (int i = 0; i < 2; i++) {
        if (rt_rows[i]) (void)hipHostFree(rt_rows[i]);
        if (rt_drows[i]) (void)hipFree(rt_drows[i]);
        if (rt_aux[i]) (void)hipFree(rt_aux[i]);
        rt_rows[i] = rt_drows[i] = rt_aux[i] = nullptr;
    }
    rt_bytes = rt_aux_bytes = 0;
}
int32_t Ctx::runtime_buffers(size_t bytes, size_t aux_bytes, uint64_t* rows[2], uint64_t* drows[2], void* aux[2], hipStream_t* copy_stream) {
    if (!rt_copy_stream) GL355_HIP(this, hipStreamCreateWithFlags(&rt_copy_stream, hipStreamNonBlocking));
    if (bytes > rt_bytes || aux_bytes > rt_aux_bytes) {
        runtime_buffers_free();
        for (int i = 0; i < 2; i++) {
            // host replay: pinned witness rows; device replay: a pinned mirror of the replay's device scratch (inputs | status | public inputs) -- the
            // side stream's copies must not start from pageable memory (the runtime stages and pins those per call: not asynchronous, and the
            // process's resident set grew by ~5 KB per unit, tools/leak_probe.py)
            hipError_t e = hipHostMalloc(&rt_rows[i], device_replay ? (aux_bytes ? aux_bytes : 256) : bytes, hipHostMallocDefault);
            if (e == hipSuccess) e = hipMalloc(&rt_drows[i], bytes);
            if (e == hipSuccess) e = hipMalloc(&rt_aux[i], aux_bytes ? aux_bytes : 256);
            if (e != hipSuccess) { (void)hipGetLastError(); runtime_buffers_free(); return fail_hip(e, "witness staging buffers", __FILE__, __LINE__); }
        }
        rt_bytes = bytes; rt_aux_bytes = aux_bytes;
    }
    for (int i = 0; i < 2; i++) { rows[i] = reinterpret_cast<uint64_t*>(rt_rows[i]); drows[i] = reinterpret_cast<uint64_t*>(rt_drows[i]); aux[i] = rt_aux[i]; }
    *copy_stream = rt_copy_stream;
    return GL355_OK;
}
void Ctx::trim_since(uint64_t mark) {
    (void)wait_impl();
    for (auto it = blocks.begin(); it != blocks.end();) {
        if (!it->used && it->serial >= mark) { (void)hipFree(it->p); it = blocks.erase(it); } else ++it;
    }
}
void Ctx::release_all() {
    for (auto& b : blocks) (void)hipFree(b.p);
    blocks.clear();
}

int32_t Ctx::pow_tables_multi(const std::vector<uint64_t>& bases, const uint64_t** lo, const uint64_t** hi) {
    auto it = pow_cache.find(bases);
    if (it != pow_cache.end()) { *lo = it->second.lo; *hi = it->second.hi; return GL355_OK; }
    const size_t nb = bases.size();
    std::vector<uint64_t> h(2 * nb * 4096);
    for (size_t c = 0; c < nb; c++) {
        uint64_t x = 1;
        const uint64_t g = bases[c];
        uint64_t* l = h.data() + c * 4096;
        for (int j = 0; j < 4096; j++) { l[j] = gl_canon(x); x = gl_mul(x, g); }
        const uint64_t g4096 = x;  // g^4096
        uint64_t* hh = h.data() + (nb + c) * 4096;
        uint64_t y = 1;
        for (int j = 0; j < 4096; j++) { hh[j] = gl_canon(y); y = gl_mul(y, g4096); }
    }
    uint64_t* d = nullptr;
    GL355_HIP(this, hipMalloc((void**)&d, h.size() * 8));
    GL355_HIP(this, hipMemcpyAsync(d, h.data(), h.size() * 8, hipMemcpyHostToDevice, stream));
    GL355_HIP(this, wait());
    PowTab t{d, d + nb * 4096};
    pow_cache[bases] = t;
    *lo = t.lo; *hi = t.hi;
    return GL355_OK;
}

bool ptr_is_device(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int32_t Staged::open(const void* ptr, size_t nbytes, int dir) {
    user = const_cast<void*>(ptr);
    bytes = nbytes;
    if (!ptr || nbytes == 0) { dev = user; is_host = false; return GL355_OK; }
    if (ptr_is_device(ptr)) { dev = user; is_host = false; return GL355_OK; }
    is_host = true;
    copy_back = (dir & 2) != 0;
    GL355_TRY(ctx->alloc(nbytes, &dev));
    if (dir & 1) GL355_HIP(ctx, hipMemcpyAsync(dev, ptr, nbytes, hipMemcpyHostToDevice, ctx->stream));
    return GL355_OK;
}
int32_t Staged::finish() {
    if (is_host && copy_back) {
        GL355_HIP(ctx, ctx->d2h(user, dev, bytes));
    }
    if (is_host) GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

static int32_t ctx_init(Ctx* c) {
    GL355_HIP(c, hipSetDevice(c->device));
    if (!c->external_stream) {
        GL355_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    GL355_HIP(c, hipEventCreate(&c->ev0));
    GL355_HIP(c, hipEventCreate(&c->ev1));
    // omega_{2^14}^(+-e) tables for the in-tile twiddles
    std::vector<uint64_t> h(2 * 16384 + 4 * 32768, 1);
    const uint64_t w = gl_root_of_unity(14), wi = gl_inv(w);
    uint64_t x = 1, y = 1;
    for (int e = 0; e < 16384; e++) { h[e] = gl_canon(x); h[16384 + e] = gl_canon(y); x = gl_mul(x, w); y = gl_mul(y, wi); }
    // the same values in the order the LDS rounds read them (Ctx::twr): [2^m + (k0 << (m - rho)) + r] = omega_{2^m}^(+-r * k0)
    for (uint32_t rho = 3; rho <= 4; rho++)
        for (uint32_t dir = 0; dir < 2; dir++) {
            uint64_t* t = h.data() + 32768 + ((rho - 3) * 2 + dir) * 32768;
            for (uint32_t m = rho + 1; m <= 14; m++)
                for (uint32_t k0 = 0; k0 < (1u << rho); k0++)
                    for (uint32_t r = 0; r < (1u << (m - rho)); r++)
                        t[(1u << m) + (k0 << (m - rho)) + r] = h[dir * 16384 + ((r * k0) << (14 - m))];
        }
    GL355_HIP(c, hipMalloc((void**)&c->tw_fwd, h.size() * 8));
    c->tw_inv = c->tw_fwd + 16384;
    GL355_HIP(c, hipMemcpyAsync(c->tw_fwd, h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream));
    GL355_HIP(c, c->wait());
    return ntt_init_constants(c);
}

}  // namespace gl355

using namespace gl355;

struct gl355_ctx {
    Ctx c;
};

// text of the last failed gl355_ctx_create on this thread (contexts are created from many threads at once)
static thread_local std::string g_create_error;

extern "C" {

const char* gl355_version(void) { return "gl355 0.1 (gfx950)"; }

int32_t gl355_device_count(int32_t* out) {
    if (!out) return GL355_E_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = 0; return GL355_E_NO_DEVICE; }
    *out = n;
    return GL355_OK;
}

static int32_t create_common(int32_t device, void* stream, bool external, gl355_ctx** out) {
    if (!out) return GL355_E_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); g_create_error = "no HIP device"; return GL355_E_NO_DEVICE; }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return GL355_E_INVALID_ARG; }
    gl355_ctx* h = new (std::nothrow) gl355_ctx();
    if (!h) return GL355_E_OOM;
    h->c.device = device;
    h->c.stream = reinterpret_cast<hipStream_t>(stream);
    h->c.own_stream = false;
    h->c.external_stream = external;
    int32_t rc = ctx_init(&h->c);
    if (rc != GL355_OK) { g_create_error = h->c.err; delete h; return rc; }
    *out = h;
    return GL355_OK;
}
int32_t gl355_runtime_config(int32_t device, uint32_t contexts, int32_t sleeping_waits) {
    if (device < 0) return GL355_E_INVALID_ARG;
    if (contexts) {
        char buf[16];
        // a context owns two streams: the proving stream and the side stream of the batch runtime (witness generation, uploads).  A
        // long interpreter kernel sharing a hardware queue with another context's proving stream would hold that stream up.
        snprintf(buf, sizeof buf, "%u", 2 * contexts < 4 ? 4u : 2 * contexts);
        setenv("GPU_MAX_HW_QUEUES", buf, 0);      // read by the HIP runtime when it initialises
    }
    if (sleeping_waits) {
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return GL355_E_HIP; }
        if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) { (void)hipGetLastError(); return GL355_E_HIP; }
    }
    return GL355_OK;
}
int32_t gl355_ctx_create(int32_t device, gl355_ctx** out) { return create_common(device, nullptr, false, out); }
// hip_stream may be NULL (the legacy default stream, which is what torch uses unless told otherwise)
int32_t gl355_ctx_create_on_stream(int32_t device, void* hip_stream, gl355_ctx** out) {
    return create_common(device, hip_stream, true, out);
}
int32_t gl355_ctx_destroy(gl355_ctx* ctx) {
    if (!ctx) return GL355_OK;
    Ctx& c = ctx->c;
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
    c.release_all();
    for (auto& kv : c.pow_cache) (void)hipFree(kv.second.lo);
    for (auto& kv : c.full_cache) (void)hipFree(kv.second);
    if (c.tw_fwd) (void)hipFree(c.tw_fwd);
    if (c.pinned_buf) (void)hipHostFree(c.pinned_buf);
    c.runtime_buffers_free();
    if (c.rt_copy_stream) (void)hipStreamDestroy(c.rt_copy_stream);
    if (c.aux_stream) (void)hipStreamDestroy(c.aux_stream);
    for (hipEvent_t e : c.order_ev) if (e) (void)hipEventDestroy(e);
    if (c.sync_ev) (void)hipEventDestroy(c.sync_ev);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    for (auto& r : c.prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : c.ev_pool) (void)hipEventDestroy(e);
    if (c.own_stream && c.stream) (void)hipStreamDestroy(c.stream);
    delete ctx;
    return GL355_OK;
}
int32_t gl355_ctx_set_option(gl355_ctx* ctx, int32_t option, int64_t value) {
    if (!ctx) return GL355_E_INVALID_ARG;
    switch (option) {
    case GL355_OPT_REPLAY_THREADS:
        if (value < 1 || value > 64) return ctx->c.fail(GL355_E_INVALID_ARG, "set_option: REPLAY_THREADS must be in 1..64");
        ctx->c.replay_threads = (uint32_t)value;
        return GL355_OK;
    case GL355_OPT_DEVICE_REPLAY:
        if (ctx->c.device_replay != (value != 0)) ctx->c.runtime_buffers_free();
        ctx->c.device_replay = value != 0;
        return GL355_OK;
    case GL355_OPT_BATCH_UNITS:
        if (value < 1 || value > GL355_MAX_UNITS) return ctx->c.fail(GL355_E_INVALID_ARG, "set_option: BATCH_UNITS must be in 1..GL355_MAX_UNITS");
        ctx->c.batch_units = (uint32_t)value;
        return GL355_OK;
    case GL355_OPT_BLOCKING_SYNC:
        if (value < 0 || value > 3) return ctx->c.fail(GL355_E_INVALID_ARG, "set_option: BLOCKING_SYNC is 0 (runtime wait), 1 (blocking event), 2 (poll + back-off) or 3 (poll, no sleep)");
        ctx->c.blocking_sync = (int)value;
        return GL355_OK;
    case GL355_OPT_NTT_SINGLE_PASS_MAX_LOG:
        if (value < 12 || value > 14) return ctx->c.fail(GL355_E_INVALID_ARG, "set_option: NTT_SINGLE_PASS_MAX_LOG must be in 12..14");
        ctx->c.ntt_single_pass_max_log = (uint32_t)value;
        return GL355_OK;
    case GL355_OPT_MERKLE_LANES_LOG:
        if (value < 0 || value > 30) return ctx->c.fail(GL355_E_INVALID_ARG, "set_option: MERKLE_LANES_LOG must be in 0..30");
        ctx->c.merkle_lanes_log = (uint32_t)value;
        return GL355_OK;
    default:
        return ctx->c.fail(GL355_E_INVALID_ARG, "set_option: unknown option");
    }
}
int32_t gl355_ctx_sync(gl355_ctx* ctx) {
    if (!ctx) return GL355_E_INVALID_ARG;
    GL355_HIP(&ctx->c, ctx->c.wait());
    return GL355_OK;
}
const char* gl355_last_error(gl355_ctx* ctx) { return ctx ? ctx->c.err.c_str() : g_create_error.c_str(); }

int32_t gl355_malloc(gl355_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return GL355_E_INVALID_ARG;
    GL355_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 256);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx->c.fail_hip(e, "hipMalloc", __FILE__, __LINE__); }
    return GL355_OK;
}
int32_t gl355_free(gl355_ctx* ctx, void* dptr) {
    if (!ctx) return GL355_E_INVALID_ARG;
    if (dptr) { GL355_HIP(&ctx->c, ctx->c.wait()); GL355_HIP(&ctx->c, hipFree(dptr)); }
    return GL355_OK;
}
int32_t gl355_memcpy_h2d(gl355_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return GL355_E_INVALID_ARG;
    GL355_HIP(&ctx->c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->c.stream));
    GL355_HIP(&ctx->c, ctx->c.wait());
    return GL355_OK;
}
int32_t gl355_memcpy_d2h(gl355_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return GL355_E_INVALID_ARG;
    GL355_HIP(&ctx->c, ctx->c.d2h(dst, src, bytes));
    GL355_HIP(&ctx->c, ctx->c.wait());
    return GL355_OK;
}
int32_t gl355_timer_start(gl355_ctx* ctx) {
    if (!ctx) return GL355_E_INVALID_ARG;
    GL355_HIP(&ctx->c, hipEventRecord(ctx->c.ev0, ctx->c.stream));
    return GL355_OK;
}
int32_t gl355_timer_stop(gl355_ctx* ctx, float* ms) {
    if (!ctx || !ms) return GL355_E_INVALID_ARG;
    GL355_HIP(&ctx->c, hipEventRecord(ctx->c.ev1, ctx->c.stream));
    GL355_HIP(&ctx->c, hipEventSynchronize(ctx->c.ev1));
    GL355_HIP(&ctx->c, hipEventElapsedTime(ms, ctx->c.ev0, ctx->c.ev1));
    return GL355_OK;
}

int32_t gl355_profile_enable(gl355_ctx* ctx, int32_t on) {
    if (!ctx) return GL355_E_INVALID_ARG;
    ctx->c.prof_on = on != 0;
    return GL355_OK;
}
// Aggregates the recorded scopes by name into `buf` as lines "name count total_ms total_algorithmic_bytes\n" and clears them.
int32_t gl355_profile_read(gl355_ctx* ctx, char* buf, size_t buf_len) {
    if (!ctx || !buf || buf_len == 0) return GL355_E_INVALID_ARG;
    Ctx& c = ctx->c;
    GL355_HIP(&c, c.wait());
    struct Agg { uint64_t n = 0; double ms = 0; uint64_t bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : c.prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { (void)hipGetLastError(); ms = 0; }
        auto& a = agg[r.name];
        a.n++; a.ms += ms; a.bytes += r.bytes;
        c.ev_pool.push_back(r.e0); c.ev_pool.push_back(r.e1);
    }
    c.prof.clear();
    std::string out;
    if (c.wait_calls) {     // pseudo-entry: calls and total wall milliseconds of Ctx::wait() while profiling was on
        out += "host:stream_wait " + std::to_string(c.wait_calls) + " " + std::to_string(c.wait_ns * 1e-6) + " 0\n";
        out += "host:cpu_in_wait " + std::to_string(c.wait_calls) + " " + std::to_string(c.wait_cpu_ns * 1e-6) + " 0\n";
        out += "host:cpu_in_prove " + std::to_string(c.prove_calls) + " " + std::to_string(c.prove_cpu_ns * 1e-6) + " 0\n";
        c.wait_calls = 0; c.wait_ns = 0; c.wait_cpu_ns = 0; c.prove_calls = 0; c.prove_cpu_ns = 0;
    }
    for (auto& kv : agg) out += kv.first + " " + std::to_string(kv.second.n) + " " + std::to_string(kv.second.ms) + " " + std::to_string(kv.second.bytes) + "\n";
    if (out.size() + 1 > buf_len) out.resize(buf_len - 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return GL355_OK;
}

}  // extern "C"

namespace gl355 {
Ctx* ctx_of(gl355_ctx* h) { return h ? &h->c : nullptr; }
}
