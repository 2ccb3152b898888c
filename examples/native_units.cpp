// Host program without any scripting layer: proves Semaphore signals and their recursive proofs through the C ABI only
// (include/gl355.h), the way a Rust binary of the reference would after INTEGRATION.md section 3b.
//
//   tools/export_artifacts.py <dir> [log_members]      writes <dir>/semaphore.gl355 and <dir>/recursive.gl355 (once per shape)
//   examples/native_units <dir> [log_members] [contexts] [units]
//   examples/native_units <dir> <log_members> <contexts> <units> --ranks <world> <rank> <id file> [--host-comm <port>]
//       one process per GPU (rank r proves on device r and takes block r of the units); the only exchange is
//       gl355_gather_digests of the (nullifier | topic) leaves, then rank 0 folds them with gl355_aggregation_root.  Rank 0 mints
//       the communicator id and writes it to <id file>, the other ranks read it from there (the "caller's own means" of
//       include/gl355.h).  --host-comm: TCP between the host processes instead of RCCL (ranks sharing one GPU, hosts without RCCL).
//
// It builds the access set like the reference does (signal.rs:31-40: public key = Poseidon(sk | 0^4), MerkleTree::new over the
// keys), loads the two circuit artifacts, runs the native batch runtime (recursion.rs:300-308) and folds the returned
// (nullifier | topic) leaves into the aggregation root.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gl355.h"

static std::vector<uint64_t> read_words(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> v(bytes / 8);
    if (fread(v.data(), 8, v.size(), f) != v.size()) { fprintf(stderr, "short read on %s\n", path.c_str()); exit(2); }
    fclose(f);
    return v;
}
static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define CHECK(ctx, expr)                                                                            \
    do {                                                                                            \
        const int32_t rc_ = (expr);                                                                 \
        if (rc_ != GL355_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, gl355_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <artifact dir> [log_members=20] [contexts=8] [units=128]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const uint32_t log_members = argc > 2 ? atoi(argv[2]) : 20, n_ctx = argc > 3 ? atoi(argv[3]) : 8;
    uint32_t units = argc > 4 ? atoi(argv[4]) : 128;
    const uint64_t n = 1ull << log_members;
    int world = 1, rank = 0, host_port = 0;
    std::string id_file;
    for (int a = 5; a < argc; a++) {
        if (!strcmp(argv[a], "--ranks") && a + 3 < argc) { world = atoi(argv[a + 1]); rank = atoi(argv[a + 2]); id_file = argv[a + 3]; a += 3; }
        else if (!strcmp(argv[a], "--host-comm") && a + 1 < argc) { host_port = atoi(argv[a + 1]); a += 1; }
    }
    int32_t n_dev = 0;
    gl355_device_count(&n_dev);
    const int device = n_dev > 0 ? rank % n_dev : 0;
    // two hardware queues per prover context (proving stream + side stream); every context waits for its stream by polling with
    // back-off (a few percent of a core per waiting context, ~30 us wake-up) and proves 8 units in lock-step
    if (gl355_runtime_config(device, n_ctx, 0) != GL355_OK) { fprintf(stderr, "gl355_runtime_config failed\n"); return 1; }
    std::vector<gl355_ctx*> ctxs(n_ctx);
    for (auto& c : ctxs) {
        if (gl355_ctx_create(device, &c) != GL355_OK) { fprintf(stderr, "no MI355X context\n"); return 1; }
        gl355_ctx_set_option(c, GL355_OPT_BLOCKING_SYNC, 2);
    }
    gl355_ctx* c0 = ctxs[0];
    // access set: secret keys from a seeded stream (values < 2^63 are canonical), public keys, Merkle tree (cap height 0)
    uint64_t seed = 0x357;
    std::vector<uint64_t> sks(4 * n), pre(8 * n, 0), keys(4 * n), digests(8 * (n - 1)), root(4), topic(4);
    for (auto& v : sks) v = splitmix(seed) >> 1;
    for (auto& v : topic) v = splitmix(seed) >> 1;
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) pre[8 * i + k] = sks[4 * i + k];
    CHECK(c0, gl355_hash_no_pad(c0, pre.data(), n, 8, keys.data()));
    CHECK(c0, gl355_merkle_build(c0, keys.data(), n, 4, 0, digests.data(), root.data()));
    // circuits
    const std::vector<uint64_t> sem_blob = read_words(dir + "/semaphore.gl355"), rec_blob = read_words(dir + "/recursive.gl355");
    gl355_circuit_handle *sem = nullptr, *rec = nullptr;
    CHECK(c0, gl355_circuit_load(c0, sem_blob.data(), sem_blob.size(), &sem));
    CHECK(c0, gl355_circuit_load(c0, rec_blob.data(), rec_blob.size(), &rec));
    uint64_t words = 0;
    uint32_t degree = 0;
    gl355_circuit_info(rec, &words, nullptr, nullptr, nullptr, &degree);
    // the communicator of the N > 1 job: id minted by rank 0, handed over through a file
    gl355_comm* comm = nullptr;
    if (world > 1) {
        uint8_t id[GL355_COMM_ID_BYTES];
        const int backend = host_port ? GL355_COMM_HOST : GL355_COMM_RCCL;
        if (rank == 0) {
            const int32_t rc = host_port ? gl355_comm_host_id("127.0.0.1", (uint16_t)host_port, id) : gl355_comm_unique_id(GL355_COMM_RCCL, id);
            if (rc != GL355_OK) { fprintf(stderr, "communicator id: %s\n", gl355_comm_last_error(nullptr)); return 1; }
            const std::string tmp = id_file + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
            fclose(f);
            rename(tmp.c_str(), id_file.c_str());
        } else {
            FILE* f = nullptr;
            for (int k = 0; k < 3000 && !(f = fopen(id_file.c_str(), "rb")); k++) std::this_thread::sleep_for(std::chrono::milliseconds(20));
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "cannot read %s\n", id_file.c_str()); return 1; }
            fclose(f);
        }
        if (gl355_comm_create(c0, backend, id, rank, world, &comm) != GL355_OK) { fprintf(stderr, "gl355_comm_create: %s\n", gl355_comm_last_error(nullptr)); return 1; }
    }
    const uint32_t all_units = units * (uint32_t)world;        // weak scaling: `units` per rank; rank r takes block r
    std::vector<uint64_t> members(units), leaves(8ull * units), proofs(words * units);
    for (uint32_t j = 0; j < units; j++) members[j] = (12 + (uint64_t)rank * units + j) % n;      // signal.rs:42 starts at index 12
    // warm-up, then the timed batch
    CHECK(c0, gl355_semaphore_units(ctxs.data(), n_ctx, sem, rec, sks.data(), n, topic.data(), digests.data(), members.data(), n_ctx, nullptr, leaves.data(),
                                    nullptr, nullptr));
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(c0, gl355_semaphore_units(ctxs.data(), n_ctx, sem, rec, sks.data(), n, topic.data(), digests.data(), members.data(), units, nullptr /* blinding keys from the OS CSPRNG */,
                                    leaves.data(), proofs.data(), nullptr));
    // the only exchange: all-gather of the leaves in rank (= unit) order, then the aggregation root on rank 0
    std::vector<uint64_t> all_leaves(8ull * all_units), aroot(4);
    if (comm) {
        if (gl355_gather_digests(comm, leaves.data(), 8ull * units, all_leaves.data()) != GL355_OK) { fprintf(stderr, "gather: %s\n", gl355_comm_last_error(comm)); return 1; }
        double dmax = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        gl355_comm_max_f64(comm, &dmax);
    } else all_leaves = leaves;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0) CHECK(c0, gl355_aggregation_root(c0, all_leaves.data(), all_units, 8, aroot.data()));
    if (comm) { gl355_comm_barrier(comm); gl355_comm_destroy(comm); }
    if (rank != 0) {
        gl355_circuit_destroy(rec); gl355_circuit_destroy(sem);
        for (auto c : ctxs) gl355_ctx_destroy(c);
        return 0;
    }
    units = all_units;
    printf("group 2^%u, %u units (signal + recursive proof, n = 2^%u) on %u contexts: %.1f units/s; %llu-word proofs\n", log_members, units, degree,
           n_ctx, units / dt, (unsigned long long)words);
    printf("access-set root %016llx..., aggregation root %016llx %016llx %016llx %016llx\n", (unsigned long long)root[0], (unsigned long long)aroot[0],
           (unsigned long long)aroot[1], (unsigned long long)aroot[2], (unsigned long long)aroot[3]);
    gl355_circuit_destroy(rec);
    gl355_circuit_destroy(sem);
    for (auto c : ctxs) gl355_ctx_destroy(c);
    return 0;
}
