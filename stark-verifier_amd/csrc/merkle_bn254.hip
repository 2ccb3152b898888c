// Second hash back-end: the reference's Bn254PoseidonHash (src/plonky2_verifier/bn245_poseidon/plonky2_config.rs:57-75
// over native.rs:16-77) for permutation batches, leaf hashing, Merkle levels, two_to_one and PoW -- the hasher of the
// final wrap proof (wrapper.rs:35-56 with OuterC = Bn254PoseidonGoldilocksConfig).  SURVEY 8(f) N1.
//
// Same launch shapes and the same plonky2 digest layout as merkle.hip; only the permutation differs (bn254.cuh).  One
// lane = one sponge state: a permutation is ~2 000 254-bit Montgomery products (~0.9 M VALU instructions, ~35x the
// Goldilocks Poseidon), so every level is throughput-bound down to a few hundred nodes and no lane-parallel variant is used.
#include "gl355_internal.h"
#include "bn254.cuh"
#include "merkle_common.cuh"

namespace gl355 {

__global__ void __launch_bounds__(256) bn254_permute_kernel(uint64_t* states, uint64_t count) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
    bn254_permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = s[k];
}

__global__ void __launch_bounds__(256) bn254_hash_leaves_kernel(LeafArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n_leaves) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    const uint32_t len = a.leaf_len;
    uint64_t d[4];
    if (len <= 4 && !a.always_hash) {
        for (uint32_t k = 0; k < 4; k++) {
            uint64_t v = 0;
            if (k < len) v = leaf_elem(a, i, k);
            d[k] = gl_canon(v);
        }
    } else {
#pragma unroll 1
        for (uint32_t off = 0; off < len; off += 8) {
            const uint32_t m = min(8u, len - off);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                if (k < m) s[k] = leaf_elem(a, i, off + k);
            }
            bn254_permute(s);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = s[k];
    }
    uint64_t* dst;
    if (a.linear) dst = a.out + i * 4;
    else if (a.sub_bits == 0) dst = a.cap + i * 4;
    else {
        const uint64_t sub_leaves = 1ull << a.sub_bits;
        const uint64_t t = i >> a.sub_bits, k = i & (sub_leaves - 1);
        dst = a.out + (t * 2 * (sub_leaves - 1) + digest_slot(0, k)) * 4;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = d[k];
}

__global__ void __launch_bounds__(256) bn254_merkle_level_kernel(uint64_t* digests, uint64_t* cap, uint32_t sub_bits, uint32_t layer,
                                                                uint64_t n_nodes) {
    const uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (g >= n_nodes) return;
    const uint64_t sub_leaves = 1ull << sub_bits;
    const uint64_t per = sub_leaves >> layer;
    const uint64_t t = g / per, k = g % per;
    uint64_t* tree = digests + t * 2 * (sub_leaves - 1) * 4;
    const uint64_t child = digest_slot(layer - 1, 2 * k);
    uint64_t s[12];
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = tree[child * 4 + e];
#pragma unroll
    for (int e = 8; e < 12; e++) s[e] = 0;
    bn254_permute(s);
    uint64_t* dst = (layer == sub_bits) ? cap + t * 4 : tree + digest_slot(layer, k) * 4;
#pragma unroll
    for (int e = 0; e < 4; e++) dst[e] = s[e];
}


// ------------------------------------------------------------------------------------------------------------------
// Lane-parallel permutation for SMALL levels.  With one lane per node a level costs one full permutation latency
// (~2.4 ms: 2 000 dependent Montgomery products) however few nodes it has, and a proof walks ~140 such levels.  Here 32
// lanes share one state as a 5 x 5 grid: lane (i, j) holds s_j (every lane of column j computes the same round-constant
// addition and S-box), forms the single product M[i][j] * s_j, and the five products of row j are summed back into s_j
// through a 25-slot LDS tile -- 4 products + 4 additions per round on the critical path instead of 28..40 products:
// ~7x lower latency at ~4.6x the instruction count, so it is used only below 2^13 nodes (BN254_LANES_MAX_LOG).
// ------------------------------------------------------------------------------------------------------------------
#define BN254_LANES_MAX_LOG 13
#define GL_WAVE_LDS_SYNC()                                      \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

GL_DEV fr8 lds_load_fr(const uint32_t* p) {
    fr8 r;
#pragma unroll
    for (int k = 0; k < FR_W; k++) r.l[k] = p[k];
    return r;
}

// block = 64 threads = two 32-lane groups = two parent nodes of layer `layer`
__global__ void __launch_bounds__(64) bn254_merkle_level_lanes_kernel(uint64_t* digests, uint64_t* cap, uint32_t sub_bits, uint32_t layer,
                                                                     uint64_t n_nodes) {
    __shared__ uint32_t tile[2][25 * FR_W];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const bool act = l < 25;
    const int i = act ? l / 5 : 0, j = act ? l % 5 : 0;
    uint64_t g = blockIdx.x * 2ull + grp;
    const bool valid = g < n_nodes;
    if (!valid) g = 0;
    const uint64_t sub_leaves = 1ull << sub_bits;
    const uint64_t per = sub_leaves >> layer;
    const uint64_t t = g / per, k = g % per;
    uint64_t* tree = digests + t * 2 * (sub_leaves - 1) * 4;
    const uint64_t child = digest_slot(layer - 1, 2 * k);
    // column j packs sponge elements 3j .. 3j+2 of (left digest | right digest | 0^4)
    uint64_t e[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int idx = 3 * j + q;
        e[q] = idx < 8 ? tree[child * 4 + idx] : 0;
    }
    fr8 s = fr_encode3(e[0], e[1], e[2]);            // column 4 (and the padding of column 2, 3) encodes zeros
    const fr8 m = fr_const(BNT(MDS)[5 * i + j]);
    uint32_t* my = tile[grp];
    fr8 rc = fr_const(BNT(RC)[j]);
#pragma unroll 1
    for (int rnd = 0; rnd < 68; rnd++) {
        const fr8 rc_next = fr_const(BNT(RC)[5 * (rnd < 67 ? rnd + 1 : 67) + j]);   // prefetch: the index depends on the lane
        s = fr_add(s, rc);
        rc = rc_next;
        const fr8 sb = fr_pow5(s);
        const bool full = rnd < 4 || rnd >= 64;
        if (full || j == 0) s = sb;
        const fr8 p = fr_mul(s, m);
        if (act) {
#pragma unroll
            for (int q = 0; q < FR_W; q++) my[(5 * i + j) * FR_W + q] = p.l[q];
        }
        GL_WAVE_LDS_SYNC();
        fr8 acc = lds_load_fr(my + (5 * j) * FR_W);     // new s_j = sum_c M[j][c] s_c: row j of the product tile
#pragma unroll
        for (int c = 1; c < 5; c++) acc = fr_add(acc, lds_load_fr(my + (5 * j + c) * FR_W));
        GL_WAVE_LDS_SYNC();
        s = acc;
    }
    uint64_t d[3];
    fr_decode3(s, d);
    // digest = sponge elements 0..3 = digits 0..2 of column 0 and digit 0 of column 1 (row 0 lanes write)
    if (valid && act && i == 0) {
        uint64_t* dst = (layer == sub_bits) ? cap + t * 4 : tree + digest_slot(layer, k) * 4;
        if (j == 0) { dst[0] = d[0]; dst[1] = d[1]; dst[2] = d[2]; }
        if (j == 1) dst[3] = d[0];
    }
}

__global__ void __launch_bounds__(256) bn254_two_to_one_kernel(const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t* out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s[12] = {gl_canon(l[4 * i]), gl_canon(l[4 * i + 1]), gl_canon(l[4 * i + 2]), gl_canon(l[4 * i + 3]),
                      gl_canon(r[4 * i]), gl_canon(r[4 * i + 1]), gl_canon(r[4 * i + 2]), gl_canon(r[4 * i + 3]), 0, 0, 0, 0};
    bn254_permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * i + k] = s[k];
}

int32_t bn254_permute_dev(Ctx* ctx, uint64_t* states, uint64_t count) {
    if (count == 0) return GL355_OK;
    ProfScope ps(ctx, "bn254_permute_kernel", count * 192);
    hipLaunchKernelGGL(bn254_permute_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, states, count);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
static int32_t launch_leaves(Ctx* ctx, const LeafArgs& a) {
    if (a.n_leaves == 0) return GL355_OK;
    ProfScope ps(ctx, "bn254_hash_leaves_kernel", a.n_leaves * ((uint64_t)a.leaf_len * 8 + 32));
    hipLaunchKernelGGL(bn254_hash_leaves_kernel, dim3((uint32_t)((a.n_leaves + 255) / 256)), dim3(256), 0, ctx->stream, a);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
int32_t bn254_hash_leaves_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major, uint64_t col_stride,
                              uint64_t* digests, bool always_hash) {
    LeafArgs a;
    memset(&a, 0, sizeof a);
    a.leaves = leaves; a.n_leaves = n_leaves; a.leaf_len = leaf_len; a.col_major = col_major;
    a.stride = col_major ? col_stride : leaf_len;
    a.out = digests; a.linear = 1; a.always_hash = always_hash;
    return launch_leaves(ctx, a);
}
int32_t bn254_two_to_one_dev(Ctx* ctx, const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t* out) {
    if (n == 0) return GL355_OK;
    hipLaunchKernelGGL(bn254_two_to_one_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, l, r, n, out);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
int32_t bn254_merkle_build_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major, uint64_t col_stride,
                               uint32_t cap_height, uint64_t* digests, uint64_t* cap) {
    const uint32_t log_n = log2_u64(n_leaves);
    if ((1ull << log_n) != n_leaves) return ctx->fail(GL355_E_INVALID_ARG, "merkle: n_leaves must be a power of two");
    if (cap_height > log_n) return ctx->fail(GL355_E_INVALID_ARG, "merkle: cap_height > log2(n_leaves)");
    LeafArgs a;
    memset(&a, 0, sizeof a);
    a.leaves = leaves; a.n_leaves = n_leaves; a.leaf_len = leaf_len; a.col_major = col_major;
    a.stride = col_major ? col_stride : leaf_len;
    return bn254_merkle_build_args(ctx, a, log_n - cap_height, digests, cap);
}
int32_t bn254_merkle_build_args(Ctx* ctx, LeafArgs a, uint32_t sub_bits, uint64_t* digests, uint64_t* cap) {
    const uint64_t n_leaves = a.n_leaves;
    if (n_leaves == 0 || (n_leaves & ((1ull << sub_bits) - 1))) return ctx->fail(GL355_E_INVALID_ARG, "merkle: leaves do not fill whole cap subtrees");
    a.out = digests; a.cap = cap; a.sub_bits = sub_bits; a.linear = 0;
    GL355_TRY(launch_leaves(ctx, a));
    for (uint32_t layer = 1; layer <= sub_bits; layer++) {
        const uint64_t n_nodes = n_leaves >> layer;
        const bool lanes = n_nodes < (1ull << BN254_LANES_MAX_LOG);
        ProfScope ps(ctx, lanes ? "bn254_merkle_level_lanes_kernel" : "bn254_merkle_level_kernel", n_nodes * 96);
        if (lanes)
            hipLaunchKernelGGL(bn254_merkle_level_lanes_kernel, dim3((uint32_t)((n_nodes + 1) / 2)), dim3(64), 0, ctx->stream, digests, cap,
                               sub_bits, layer, n_nodes);
        else
            hipLaunchKernelGGL(bn254_merkle_level_kernel, dim3((uint32_t)((n_nodes + 255) / 256)), dim3(256), 0, ctx->stream, digests, cap,
                               sub_bits, layer, n_nodes);
        GL355_HIP(ctx, hipGetLastError());
    }
    return GL355_OK;
}

// proof of work with this hasher: same search as pow_grind_dev (smallest passing candidate of the first launch that holds one)
__global__ void __launch_bounds__(256) bn254_pow_grind_kernel(const uint64_t* state, uint32_t pos, uint32_t bits, uint64_t start,
                                                             unsigned long long* best) {
    const uint64_t w = start + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = state[k];
#pragma unroll
    for (int k = 0; k < 12; k++) if ((uint32_t)k == pos) s[k] = w;
    bn254_permute(s);
    if (bits == 0 || (s[7] >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)w);
}
int32_t bn254_pow_grind_dev(Ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start, uint64_t* witness_host) {
    Scratch sc(ctx);
    GL355_TRY(sc.get(13 * sizeof(uint64_t)));
    uint64_t* d_state = sc.as<uint64_t>();
    unsigned long long* d_best = reinterpret_cast<unsigned long long*>(d_state + 12);
    uint64_t host[13];
    for (int i = 0; i < 12; i++) host[i] = gl_canon(state[i]);
    host[12] = ~0ull;
    GL355_HIP(ctx, hipMemcpyAsync(d_state, host, sizeof host, hipMemcpyHostToDevice, ctx->stream));
    uint64_t per_launch = 1ull << std::min<uint32_t>(std::max<uint32_t>(bits + 1, 12), 20);
    uint64_t base = start;
    for (;;) {
        ProfScope ps(ctx, "bn254_pow_grind", 0);
        hipLaunchKernelGGL(bn254_pow_grind_kernel, dim3((uint32_t)(per_launch / 256)), dim3(256), 0, ctx->stream, d_state, pos, bits, base,
                           d_best);
        GL355_HIP(ctx, hipGetLastError());
        unsigned long long best;
        GL355_HIP(ctx, ctx->d2h(&best, d_best, sizeof best));
        GL355_HIP(ctx, ctx->wait());
        if (best != ~0ull) { *witness_host = best; return GL355_OK; }
        base += per_launch;
        if (per_launch < (1ull << 20)) per_launch <<= 1;
        if (base - start > (1ull << 40)) return ctx->fail(GL355_E_UNSUPPORTED, "pow: no witness found in 2^40 candidates");
    }
}

int32_t merkle_build_any(Ctx* ctx, int32_t hasher, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major,
                         uint64_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap) {
    if (hasher == GL355_HASH_BN254_POSEIDON)
        return bn254_merkle_build_dev(ctx, leaves, n_leaves, leaf_len, col_major, col_stride, cap_height, digests, cap);
    return merkle_build_dev(ctx, leaves, n_leaves, leaf_len, col_major, col_stride, cap_height, digests, cap);
}
int32_t merkle_build_args_any(Ctx* ctx, int32_t hasher, const LeafArgs& a, uint32_t sub_bits, uint64_t* digests, uint64_t* cap) {
    if (hasher == GL355_HASH_BN254_POSEIDON) return bn254_merkle_build_args(ctx, a, sub_bits, digests, cap);
    return merkle_build_args(ctx, a, sub_bits, digests, cap);
}
int32_t pow_grind_any(Ctx* ctx, int32_t hasher, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start,
                      uint64_t* witness_host) {
    if (hasher == GL355_HASH_BN254_POSEIDON) return bn254_pow_grind_dev(ctx, state, pos, bits, start, witness_host);
    return pow_grind_dev(ctx, state, pos, bits, start, witness_host);
}

}  // namespace gl355
