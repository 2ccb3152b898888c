// Multi-GPU exchange behind the C ABI (SURVEY 8(b) `gl355_gather_digests`, 8(e)).
//
// The reference collects the results of its independent proofs from a rayon `par_iter` into a `Mutex<Vec<Signal>>`
// (src/plonky2_semaphore/recursion.rs:189-227, 300-308).  With one process per GPU that collection is the ONLY exchange of the
// workload: an all-gather of one small leaf per unit (nullifier | topic = 8 words, recursion.rs:110-165) -- or of one proof per
// rank for the aggregation tree -- after which rank 0 builds the aggregation root with the HIP Merkle kernel.  No data-path
// collective exists besides it, so this file is small on purpose:
//   GL355_COMM_RCCL  ncclAllGather / ncclAllReduce over xGMI on the context's stream.  The communicator is created from a unique
//                    id the CALLER distributes (rank 0 mints it with gl355_comm_unique_id and hands the 128 bytes to the other
//                    ranks by the host's own means: its launcher, a file, a socket) -- the library opens no side channel.
//                    librccl is bound at the first use (dlopen), so a single-GPU host needs no RCCL installation.
//   GL355_COMM_HOST  the same calls over TCP between the host processes (rank 0 listens on the address the id names): hosts
//                    without RCCL, and the communicator of the CPU test-suite.  Chosen explicitly, never as a fallback.
// Messages are a few KB (64 B per unit; ~0.2 MB per rank for the proof exchange): latency-bound, a ring or a tree makes no
// difference, so the HOST backend is a star through rank 0.
#include "gl355_internal.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <fcntl.h>
#include <poll.h>
#include <sys/random.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>

using namespace gl355;

namespace {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool ok = false;
};

// the process's RCCL: the copy already loaded (torch ships one under the same soname) or the system one
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { api.error = std::string("librccl not found: ") + dlerror(); return; }
#define GL355_SYM(field, name)                                                        \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));             \
        if (!api.field) { api.error = std::string("librccl lacks ") + name; return; }
        GL355_SYM(GetUniqueId, "ncclGetUniqueId")
        GL355_SYM(CommInitRank, "ncclCommInitRank")
        GL355_SYM(CommDestroy, "ncclCommDestroy")
        GL355_SYM(AllGather, "ncclAllGather")
        GL355_SYM(AllReduce, "ncclAllReduce")
        GL355_SYM(GetErrorString, "ncclGetErrorString")
#undef GL355_SYM
        api.ok = true;
    });
    return api;
}

bool send_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; }
        c += w; n -= (size_t)w;
    }
    return true;
}
bool recv_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t r = ::recv(fd, c, n, 0);
        if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
        c += r; n -= (size_t)r;
    }
    return true;
}

constexpr char HOST_ID_MAGIC[8] = {'g', 'l', '3', '5', '5', 't', 'c', 'p'};
constexpr size_t HOST_TOKEN_OFFSET = 16, HOST_TOKEN_BYTES = 16;       // id bytes [16, 32): the handshake token

// seconds a data-socket send / receive may block before the call fails (a dead peer must not hang every rank for ever)
int comm_timeout_s() {
    const char* e = getenv("GL355_COMM_TIMEOUT_S");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 600;
}
void set_io_timeouts(int fd, int seconds) {
    const timeval t{seconds, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &t, sizeof t);
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &t, sizeof t);
}

}  // namespace

struct gl355_comm {
    int32_t backend = GL355_COMM_RCCL;
    int32_t rank = 0, world = 1;
    Ctx* ctx = nullptr;                 // RCCL: stream + staging buffers; HOST: optional (device operands)
    ncclComm_t nccl = nullptr;
    std::vector<int> peers;             // HOST, rank 0: socket of every rank (index = rank; [0] unused); other ranks: [0] = rank 0
    int listen_fd = -1;
    std::string err;
    int32_t fail(int32_t code, const std::string& m) { err = m; if (ctx) ctx->err = m; return code; }
};

static thread_local std::string g_comm_error;

extern "C" {

const char* gl355_comm_last_error(gl355_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int32_t gl355_comm_unique_id(int32_t backend, uint8_t id[GL355_COMM_ID_BYTES]) {
    if (!id) return GL355_E_INVALID_ARG;
    if (backend != GL355_COMM_RCCL) { g_comm_error = "comm_unique_id: HOST ids are built with gl355_comm_host_id"; return GL355_E_INVALID_ARG; }
    RcclApi& r = rccl();
    if (!r.ok) { g_comm_error = r.error; return GL355_E_UNSUPPORTED; }
    static_assert(sizeof(ncclUniqueId) == GL355_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    const ncclResult_t rc = r.GetUniqueId(&u);
    if (rc != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return GL355_E_HIP; }
    memcpy(id, &u, sizeof u);
    return GL355_OK;
}

int32_t gl355_comm_host_id(const char* ipv4, uint16_t port, uint8_t id[GL355_COMM_ID_BYTES]) {
    if (!ipv4 || !id || port == 0) return GL355_E_INVALID_ARG;
    in_addr a;
    if (inet_pton(AF_INET, ipv4, &a) != 1) { g_comm_error = "comm_host_id: not a dotted IPv4 address"; return GL355_E_INVALID_ARG; }
    memset(id, 0, GL355_COMM_ID_BYTES);
    memcpy(id, HOST_ID_MAGIC, 8);
    memcpy(id + 8, &a, 4);
    id[12] = (uint8_t)(port & 0xFF); id[13] = (uint8_t)(port >> 8);
    // a random token every rank must present in its handshake: the id is minted ONCE (by rank 0) and handed to the other ranks by
    // the host's launcher, like an ncclUniqueId -- a process that merely reaches the port cannot claim a rank
    size_t got = 0;
    while (got < HOST_TOKEN_BYTES) {
        const ssize_t r = getrandom(id + HOST_TOKEN_OFFSET + got, HOST_TOKEN_BYTES - got, 0);
        if (r < 0) { if (errno == EINTR) continue; g_comm_error = "comm_host_id: getrandom failed"; return GL355_E_UNSUPPORTED; }
        got += (size_t)r;
    }
    return GL355_OK;
}

int32_t gl355_comm_create(gl355_ctx* h, int32_t backend, const uint8_t id[GL355_COMM_ID_BYTES], int32_t rank, int32_t world, gl355_comm** out) {
    if (!out) return GL355_E_INVALID_ARG;
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) { g_comm_error = "comm_create: bad rank / world / id"; return GL355_E_INVALID_ARG; }
    Ctx* ctx = ctx_of(h);
    gl355_comm* c = new (std::nothrow) gl355_comm();
    if (!c) return GL355_E_OOM;
    c->backend = backend; c->rank = rank; c->world = world; c->ctx = ctx;
    if (backend == GL355_COMM_RCCL) {
        if (!ctx) { delete c; g_comm_error = "comm_create: the RCCL backend needs a context (device + stream)"; return GL355_E_INVALID_ARG; }
        RcclApi& r = rccl();
        if (!r.ok) { g_comm_error = r.error; ctx->err = r.error; delete c; return GL355_E_UNSUPPORTED; }
        if (hipSetDevice(ctx->device) != hipSuccess) { delete c; return ctx->fail(GL355_E_HIP, "hipSetDevice failed"); }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        const ncclResult_t rc = r.CommInitRank(&c->nccl, world, u, rank);
        if (rc != ncclSuccess) {
            g_comm_error = std::string("ncclCommInitRank: ") + r.GetErrorString(rc);
            ctx->err = g_comm_error;
            delete c;
            return GL355_E_HIP;
        }
        *out = c;
        return GL355_OK;
    }
    if (backend != GL355_COMM_HOST) { delete c; g_comm_error = "comm_create: unknown backend"; return GL355_E_INVALID_ARG; }
    if (memcmp(id, HOST_ID_MAGIC, 8) != 0) { delete c; g_comm_error = "comm_create: not a gl355_comm_host_id id"; return GL355_E_INVALID_ARG; }
    sockaddr_in addr;
    memset(&addr, 0, sizeof addr);
    addr.sin_family = AF_INET;
    memcpy(&addr.sin_addr, id + 8, 4);
    addr.sin_port = htons((uint16_t)(id[12] | (id[13] << 8)));
    const int one = 1;
    if (world == 1) { *out = c; return GL355_OK; }
    if (rank == 0) {
        c->peers.assign(world, -1);
        c->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (c->listen_fd < 0) { delete c; g_comm_error = "comm_create: socket()"; return GL355_E_UNSUPPORTED; }
        setsockopt(c->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (::bind(c->listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) != 0 || ::listen(c->listen_fd, world) != 0) {
            g_comm_error = std::string("comm_create: cannot listen on the id's address: ") + strerror(errno);
            ::close(c->listen_fd); delete c;
            return GL355_E_UNSUPPORTED;
        }
        // Accept until every rank has arrived or the ONE deadline passes.  Accepted sockets are non-blocking and polled together with the
        // listener: a connection that stays silent (a port scan, a stray client) costs nothing but its slot until it is dropped after 5 s,
        // and cannot hold real ranks back in the backlog.  A connection whose (rank, token) is complete and valid gets a 1-byte
        // acknowledgement and claims its rank; anything else -- wrong token, duplicate or out-of-range rank -- is closed, which the
        // peer sees at once (gl355_comm_create fails there instead of at its first gather).
        struct Handshake { int32_t rank; uint8_t token[HOST_TOKEN_BYTES]; };
        struct Pending { int fd; size_t got; Handshake hs; std::chrono::steady_clock::time_point since; };
        std::vector<Pending> pending;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        int have = 1;
        while (have < world) {
            const auto now = std::chrono::steady_clock::now();
            const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - now).count();
            if (left <= 0) break;
            for (size_t i = 0; i < pending.size();) {                 // silent for 5 s: dropped
                if (now - pending[i].since > std::chrono::seconds(5)) { ::close(pending[i].fd); pending.erase(pending.begin() + i); }
                else i++;
            }
            std::vector<pollfd> pfs(1 + pending.size());
            pfs[0] = pollfd{c->listen_fd, POLLIN, 0};
            for (size_t i = 0; i < pending.size(); i++) pfs[1 + i] = pollfd{pending[i].fd, POLLIN, 0};
            const int pr = ::poll(pfs.data(), (nfds_t)pfs.size(), (int)std::min<long long>(left, 250));
            if (pr < 0 && errno != EINTR) break;
            if (pr <= 0) continue;
            const size_t n_pending = pending.size();                  // sockets accepted below are polled on the next turn
            if (pfs[0].revents & POLLIN) {
                const int fd = ::accept4(c->listen_fd, nullptr, nullptr, SOCK_NONBLOCK);
                if (fd >= 0) {
                    if (pending.size() >= 1024) ::close(fd);          // bounded: a flood cannot exhaust descriptors
                    else pending.push_back(Pending{fd, 0, Handshake{}, std::chrono::steady_clock::now()});
                }
            }
            std::vector<size_t> done;
            for (size_t i = 0; i < n_pending; i++) {
                if (!(pfs[1 + i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
                Pending& pe = pending[i];
                const ssize_t r = ::recv(pe.fd, reinterpret_cast<char*>(&pe.hs) + pe.got, sizeof(Handshake) - pe.got, 0);
                if (r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) continue;
                if (r <= 0) { ::close(pe.fd); pe.fd = -1; done.push_back(i); continue; }
                pe.got += (size_t)r;
                if (pe.got < sizeof(Handshake)) continue;
                const bool ok = pe.hs.rank > 0 && pe.hs.rank < world && c->peers[pe.hs.rank] < 0 &&
                                memcmp(pe.hs.token, id + HOST_TOKEN_OFFSET, HOST_TOKEN_BYTES) == 0;
                if (ok) {
                    const int fl = fcntl(pe.fd, F_GETFL, 0);
                    if (fl >= 0) (void)fcntl(pe.fd, F_SETFL, fl & ~O_NONBLOCK);
                    setsockopt(pe.fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                    set_io_timeouts(pe.fd, comm_timeout_s());
                    const uint8_t ack = 1;
                    if (send_all(pe.fd, &ack, 1)) { c->peers[pe.hs.rank] = pe.fd; have++; pe.fd = -1; }
                }
                if (pe.fd >= 0) ::close(pe.fd);
                pe.fd = -1;
                done.push_back(i);
            }
            for (size_t k = done.size(); k-- > 0;) pending.erase(pending.begin() + done[k]);
        }
        for (const Pending& pe : pending) if (pe.fd >= 0) ::close(pe.fd);
        if (have < world) {
            gl355_comm_destroy(c);
            g_comm_error = "comm_create: not every rank presented a valid handshake before the deadline";
            return GL355_E_UNSUPPORTED;
        }
    } else {
        int fd = -1;
        for (int attempt = 0; attempt < 600; attempt++) {          // rank 0 may not be listening yet: retry for ~60 s
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && ::connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) == 0) break;
            if (fd >= 0) ::close(fd);
            fd = -1;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        struct { int32_t rank; uint8_t token[HOST_TOKEN_BYTES]; } hs;
        hs.rank = rank;
        memcpy(hs.token, id + HOST_TOKEN_OFFSET, HOST_TOKEN_BYTES);
        if (fd < 0 || !send_all(fd, &hs, sizeof hs)) {
            if (fd >= 0) ::close(fd);
            delete c;
            g_comm_error = "comm_create: cannot reach rank 0";
            return GL355_E_UNSUPPORTED;
        }
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        // rank 0 acknowledges a valid handshake with one byte and closes a rejected one: a stale id or token fails here, not at the first gather
        set_io_timeouts(fd, 125);
        uint8_t ack = 0;
        if (!recv_all(fd, &ack, 1) || ack != 1) {
            ::close(fd);
            delete c;
            g_comm_error = "comm_create: rank 0 rejected the handshake (stale id or token, duplicate rank) or never answered";
            return GL355_E_UNSUPPORTED;
        }
        set_io_timeouts(fd, comm_timeout_s());
        c->peers.assign(1, fd);
    }
    *out = c;
    return GL355_OK;
}

int32_t gl355_comm_destroy(gl355_comm* c) {
    if (!c) return GL355_OK;
    if (c->nccl) { if (c->ctx) (void)hipSetDevice(c->ctx->device); (void)rccl().CommDestroy(c->nccl); }
    for (int fd : c->peers) if (fd >= 0) ::close(fd);
    if (c->listen_fd >= 0) ::close(c->listen_fd);
    delete c;
    return GL355_OK;
}

int32_t gl355_comm_info(const gl355_comm* c, int32_t* rank, int32_t* world, int32_t* backend) {
    if (!c) return GL355_E_INVALID_ARG;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (backend) *backend = c->backend;
    return GL355_OK;
}

// host-side star all-gather of `bytes` per rank (HOST backend)
static int32_t host_allgather(gl355_comm* c, const void* local, size_t bytes, void* all) {
    char* dst = static_cast<char*>(all);
    if (c->world == 1) { memcpy(dst, local, bytes); return GL355_OK; }
    if (c->rank == 0) {
        memcpy(dst, local, bytes);
        for (int k = 1; k < c->world; k++)
            if (!recv_all(c->peers[k], dst + (size_t)k * bytes, bytes)) return c->fail(GL355_E_HIP, "comm: a peer closed its connection");
        for (int k = 1; k < c->world; k++)
            if (!send_all(c->peers[k], dst, bytes * c->world)) return c->fail(GL355_E_HIP, "comm: a peer closed its connection");
    } else {
        if (!send_all(c->peers[0], local, bytes) || !recv_all(c->peers[0], dst, bytes * c->world)) return c->fail(GL355_E_HIP, "comm: rank 0 closed its connection");
    }
    return GL355_OK;
}

int32_t gl355_gather_digests(gl355_comm* c, const uint64_t* local, uint64_t words_per_rank, uint64_t* all) {
    if (!c) return GL355_E_INVALID_ARG;
    if (words_per_rank == 0) return GL355_OK;
    if (!local || !all) return c->fail(GL355_E_INVALID_ARG, "gather_digests: null buffer");
    if (words_per_rank > (UINT64_MAX / 8) / (uint64_t)c->world) return c->fail(GL355_E_INVALID_ARG, "gather_digests: words_per_rank * world overflows");
    const size_t bytes = (size_t)words_per_rank * 8;
    if (c->backend == GL355_COMM_HOST) {
        const bool ld = ptr_is_device(local), ad = ptr_is_device(all);
        if (!ld && !ad) return host_allgather(c, local, bytes, all);
        // device operands: stage through the host
        std::vector<uint64_t> lh(words_per_rank), ah((size_t)words_per_rank * c->world);
        if (ld) { if (hipMemcpy(lh.data(), local, bytes, hipMemcpyDeviceToHost) != hipSuccess) return c->fail(GL355_E_HIP, "gather_digests: copy from the device"); }
        else memcpy(lh.data(), local, bytes);
        GL355_TRY(host_allgather(c, lh.data(), bytes, ah.data()));
        if (ad) { if (hipMemcpy(all, ah.data(), bytes * c->world, hipMemcpyHostToDevice) != hipSuccess) return c->fail(GL355_E_HIP, "gather_digests: copy to the device"); }
        else memcpy(all, ah.data(), bytes * c->world);
        return GL355_OK;
    }
    Ctx* ctx = c->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    Staged sl(ctx), sa(ctx);
    GL355_TRY(sl.open(local, bytes, 1));
    GL355_TRY(sa.open(all, bytes * c->world, 2));
    const ncclResult_t rc = rccl().AllGather(sl.as<void>(), sa.as<void>(), (size_t)words_per_rank, ncclUint64, c->nccl, ctx->stream);
    if (rc != ncclSuccess) return c->fail(GL355_E_HIP, std::string("ncclAllGather: ") + rccl().GetErrorString(rc));
    GL355_TRY(sa.finish());
    GL355_HIP(ctx, ctx->wait());     // device operands: the caller's `local` may be reused on return
    return GL355_OK;
}

static int32_t reduce_max_f64(gl355_comm* c, double* v) {
    if (c->backend == GL355_COMM_HOST) {
        std::vector<double> all(c->world);
        GL355_TRY(host_allgather(c, v, sizeof(double), all.data()));
        for (double x : all) if (x > *v) *v = x;
        return GL355_OK;
    }
    Ctx* ctx = c->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    Scratch sc(ctx);
    GL355_TRY(sc.get(16));
    GL355_HIP(ctx, hipMemcpyAsync(sc.p, v, 8, hipMemcpyHostToDevice, ctx->stream));
    const ncclResult_t rc = rccl().AllReduce(sc.p, sc.p, 1, ncclFloat64, ncclMax, c->nccl, ctx->stream);
    if (rc != ncclSuccess) return c->fail(GL355_E_HIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(rc));
    GL355_HIP(ctx, ctx->d2h(v, sc.p, 8));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

int32_t gl355_comm_max_f64(gl355_comm* c, double* inout) {
    if (!c || !inout) return GL355_E_INVALID_ARG;
    return reduce_max_f64(c, inout);
}

int32_t gl355_comm_barrier(gl355_comm* c) {
    if (!c) return GL355_E_INVALID_ARG;
    double z = 0;
    return reduce_max_f64(c, &z);
}

// "aggregation root" of the gathered leaves: Poseidon-Goldilocks Merkle root (cap height 0) over n_leaves leaves of
// leaf_len words, zero-padded to a power of two -- MerkleTree::new(leaves, 0).cap[0] (recursion.rs:360 builds the same tree shape)
int32_t gl355_aggregation_root(gl355_ctx* h, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, uint64_t root[4]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!leaves || !root || n_leaves == 0 || leaf_len == 0) return ctx->fail(GL355_E_INVALID_ARG, "aggregation_root: empty input");
    uint64_t m = 1;
    while (m < n_leaves) m <<= 1;
    Scratch buf(ctx);
    GL355_TRY(buf.get((m * leaf_len + 2 * (m - 1) * 4 + 4 + 8) * 8));
    uint64_t* d_leaves = buf.as<uint64_t>();
    uint64_t* d_dig = d_leaves + m * leaf_len;
    uint64_t* d_cap = d_dig + 2 * (m - 1) * 4;
    GL355_HIP(ctx, hipMemsetAsync(d_leaves, 0, m * leaf_len * 8, ctx->stream));
    GL355_HIP(ctx, hipMemcpyAsync(d_leaves, leaves, n_leaves * leaf_len * 8, ptr_is_device(leaves) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    GL355_TRY(merkle_build_dev(ctx, d_leaves, m, leaf_len, false, 0, 0, d_dig, d_cap));
    if (ptr_is_device(root)) GL355_HIP(ctx, hipMemcpyAsync(root, d_cap, 32, hipMemcpyDeviceToDevice, ctx->stream));
    else GL355_HIP(ctx, ctx->d2h(root, d_cap, 32));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

}  // extern "C"
