// Poseidon sponge, Merkle tree with cap and proof-of-work grinding for gfx950 (a6, a7, a8, a13).
//
// Replaces plonky2::hash::hashing::{hash_n_to_m_no_pad, compress}, Hasher::{hash_or_noop,
// two_to_one}, plonky2::hash::merkle_tree::{MerkleTree::new, fill_digests_buf, fill_subtree} and
// plonky2::fri::prover::fri_proof_of_work; reached from the reference at
// src/plonky2_semaphore/signal.rs:40, access_set.rs:67,205, recursion.rs:360 and inside every
// PolynomialBatch commitment.  Semantics pinned by chip/hasher_chip.rs:122-148 (overwrite sponge,
// rate 8), chip/merkle_proof_chip.rs:39-87 (leaf <= 4 elements is its own digest; bit k of the
// index = 1 means "current node is the right child") and chip/fri_chip.rs:364-376 (PoW).
//
// Design: these kernels are integer-VALU bound (one permutation ~ 1.1 k 64-bit modmuls), not HBM
// bound, so the layout goal is only "never waste a load": leaves are read straight from the
// column-major LDE the NTT wrote (lane i reads element i of a column: perfectly coalesced; the
// row-major transpose plonky2 performs is never materialised on the commit path), and the digests
// are written directly into plonky2's recursive `digests` layout so no later shuffle is needed.
#include "gl355_internal.h"
#include "poseidon.cuh"
#include "merkle_common.cuh"

namespace gl355 {

// levels with at most this many nodes use the lane-parallel kernel (4 nodes per wave): beyond ~2^14 nodes
// the chip is full of waves either way and the one-lane-per-node kernel wins on instruction count
// the default of Ctx::merkle_lanes_log (gl355_ctx_set_option) is 14: see the lane-parallel kernel below

__global__ void __launch_bounds__(256) poseidon_permute_kernel(uint64_t* states, uint64_t count) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
    psd_permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = gl_canon(s[k]);
}


// one lane = one leaf: overwrite-mode sponge over ceil(len/8) chunks
#ifndef PSD_LEAF_WAVES
#define PSD_LEAF_WAVES 4
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PSD_LEAF_WAVES))) hash_leaves_kernel(LeafArgs a) {
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
        for (uint32_t off = 0; off < len; off += 8) {
            const uint32_t m = min(8u, len - off);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                if (k < m) s[k] = leaf_elem(a, i, off + k);
            }
            psd_permute(s);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = gl_canon(s[k]);
    }
    uint64_t* dst;
    if (a.linear) dst = a.out + i * 4;
    else if (a.sub_bits == 0) dst = a.cap + i * 4;
    else {
        const uint64_t sub_leaves = 1ull << a.sub_bits;
        const uint64_t t = i >> a.sub_bits, k = i & (sub_leaves - 1);
        dst = a.out + (t * 2 * (sub_leaves - 1) + digest_slot(0, k)) * 4;
    }
    *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(d[0], d[1]);
    *reinterpret_cast<ulonglong2*>(dst + 2) = make_ulonglong2(d[2], d[3]);
}

// one lane = one parent node of layer `layer` (>= 1): two_to_one of its children in layer-1
__global__ void __launch_bounds__(256) merkle_level_kernel(uint64_t* digests, uint64_t* cap, uint32_t sub_bits,
                                                          uint32_t layer, uint64_t n_nodes /* all subtrees */) {
    const uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (g >= n_nodes) return;
    const uint64_t sub_leaves = 1ull << sub_bits;
    const uint64_t per = sub_leaves >> layer;
    const uint64_t t = g / per, k = g % per;
    uint64_t* tree = digests + t * 2 * (sub_leaves - 1) * 4;
    const uint64_t child = digest_slot(layer - 1, 2 * k);  // left child; right child is +1
    const ulonglong2 l0 = *reinterpret_cast<const ulonglong2*>(tree + child * 4);
    const ulonglong2 l1 = *reinterpret_cast<const ulonglong2*>(tree + child * 4 + 2);
    const ulonglong2 r0 = *reinterpret_cast<const ulonglong2*>(tree + child * 4 + 4);
    const ulonglong2 r1 = *reinterpret_cast<const ulonglong2*>(tree + child * 4 + 6);
    uint64_t s[12] = {l0.x, l0.y, l1.x, l1.y, r0.x, r0.y, r1.x, r1.y, 0, 0, 0, 0};
    psd_permute(s);
    uint64_t* dst = (layer == sub_bits) ? cap + t * 4 : tree + digest_slot(layer, k) * 4;
    *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
    *reinterpret_cast<ulonglong2*>(dst + 2) = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}

// ------------------------------------------------------------------------------------------------
// Lane-parallel permutation for SMALL levels.  A thread-per-node launch of a level with few nodes is
// one lonely wave per SIMD running a ~26 k-instruction dependent stream (~75 us) no matter how few
// nodes there are, and a proof walks ~100 such levels (profiles/r01_proof_kernel_stats.csv: 61 % of
// GPU time).  Here 16 lanes share one state (lane i < 12 holds element i): the S-box is one x^7 per
// lane, and the MDS row of lane r is read from a 24-slot LDS ring (state written twice, so slot
// r + j needs no modulo) -- ~10x fewer instructions per wave, i.e. ~10x lower level latency.  All
// 30 rounds use the naive form (constants + S-box + full MDS; the S-box result is kept on lane 0
// only in the 22 partial rounds): with one element per lane the dense MDS is as cheap as the sparse one.
// ------------------------------------------------------------------------------------------------
__device__ __constant__ const uint32_t PSD_CIRC_DEV[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

// the 16 lanes sharing a state sit in ONE wave, and a wave's LDS instructions complete in issue order: the ring
// exchange only needs the compiler (and the LGKM counter) to keep write -> read -> next write ordered, not s_barrier
#define GL_WAVE_LDS_SYNC()                                      \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
// PSD_LANES_FORM (round 6): the 22 PARTIAL rounds of the 16-lane permutation.
//   0  as the full rounds: S-box (only lane 0's matters), the state through the ring, the whole MDS row: S-box + LDS round trip + 24 multiply-adds +
//      recombination, one after the other on the critical path;
//   3  only element 0 changes in a partial round's S-box, so the other eleven elements go through the ring BEFORE it (lane 0 publishes a zero), and each
//      lane's row over them is accumulated INSIDE lane 0's x^7: the row's twelve terms (two multiply-adds each) are the fillers of the products' carry
//      wait states (gl_mul_fill / gl_mul2_fill) -- written any other way the compiler queues the row's multiply-adds as a block in front of the S-box
//      chain (forms 1 and 2 of profiles/r06_lanes_partial_rounds_ab.txt: no gain).  x^7 then arrives by a DPP row broadcast (row_newbcast:0: one 16-lane
//      row = one state) for the last two multiply-adds.
// Form 3 is 16 % faster per permutation (tree tops 0.080 -> 0.067 ms, a lone unit 11.7 -> 11.0 ms) and costs 16 more VGPRs: under a lock-step load, where
// these kernels share the CUs with the hash kernels of nine other contexts, the job measured 0.8 % SLOWER with it (320.8 vs 323.3 units/s) -- so the
// launch code takes form 3 for a lone proof's forest (fewer than 64 cap subtrees) and form 0 for lock-step batches.  PSD_LANES_FORM = 0 / 3 forces one (A/B).
#ifndef PSD_LANES_FORM
#define PSD_LANES_FORM -1
#endif
template <int FORM>
GL_DEV uint64_t psd_permute_lanes(uint64_t s, int li, uint64_t* ring /* 24 u64 of this 16-lane group */) {
    const bool active = li < 12;
    const int me = active ? li : 0;
    // the constant index depends on the lane, so this is a vector load: fetch round r+1's constant while round r
    // computes (a load issued and awaited inside the round costs an L2 round trip per round, ~40 % of the latency)
    s = gl_add_canonical(s, PSD_ALL_RC[me]);
    auto dense_round = [&](int r, bool full) {
        const uint64_t rc_next = PSD_ALL_RC[12 * (r + 1) + me];     // row 30 is zero; consumed by this round's MDS accumulators
        if (full || li == 0) s = psd_sbox(s);
        if (active) { ring[me] = s; ring[me + 12] = s; }
        GL_WAVE_LDS_SYNC();
        uint64_t al = (uint32_t)rc_next, ah = rc_next >> 32;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint64_t x = ring[me + j];
            const uint32_t c = PSD_CIRC_DEV[j] + ((me == 0 && j == 0) ? 8u : 0u);
            al += (uint64_t)(uint32_t)x * c;
            ah += (uint64_t)(uint32_t)(x >> 32) * c;
        }
        GL_WAVE_LDS_SYNC();
        s = psd_recombine(al, ah);
    };
#if !defined(__HIP_DEVICE_COMPILE__) || GL_MUL_VARIANT != 1
#pragma unroll 1
    for (int r = 0; r < 30; r++) dense_round(r, r < 4 || r >= 26);
#else
    if constexpr (FORM == 0) {
#pragma unroll 1
        for (int r = 0; r < 30; r++) dense_round(r, r < 4 || r >= 26);
        return s;
    }
    // row `me`'s coefficient of element 0: M[me][0] = CIRC[(12 - me) % 12], + 8 on the diagonal
    const uint32_t c_elem0 = PSD_CIRC_DEV[(12 - me) % 12] + (me == 0 ? 8u : 0u);
#pragma unroll 1
    for (int r = 0; r < 4; r++) dense_round(r, true);
#pragma unroll 1
    for (int r = 4; r < 26; r++) {
        const uint64_t rc_next = PSD_ALL_RC[12 * (r + 1) + me];
        if (active) { const uint64_t pub = li == 0 ? 0 : s; ring[me] = pub; ring[me + 12] = pub; }     // element 0 is added below, after its S-box
        GL_WAVE_LDS_SYNC();
        uint64_t xr[12];
#pragma unroll
        for (int j = 0; j < 12; j++) xr[j] = ring[me + j];
        uint64_t al = (uint32_t)rc_next, ah = rc_next >> 32;
        auto term = [&](int j) {                                   // one element of the row: two multiply-adds (the + 8 of row 0 belongs to element 0)
            al += (uint64_t)(uint32_t)xr[j] * PSD_CIRC_DEV[j];
            ah += (uint64_t)(uint32_t)(xr[j] >> 32) * PSD_CIRC_DEV[j];
        };
        // x^7 = (x x^2)(x^2 x^2), every lane computes it (lane 0's is the one used); terms 0 .. 11 ride in the carry wait states of its four products
        const uint64_t x2 = gl_mul_fill(s, s, [&](int k) { term(k); });
        uint64_t p34[2];
        { const uint64_t pa[2] = {s, x2}, pb[2] = {x2, x2}; gl_mul2_fill(pa, pb, p34, [&](int k) { term(5 + k); }); }
        const uint64_t x7 = gl_mul_fill(p34[0], p34[1], [&](int k) { if (k < 2) term(10 + k); else asm volatile("s_nop 1"); });
        const uint32_t x0l = __builtin_amdgcn_update_dpp(0u, (uint32_t)x7, 0x150, 0xf, 0xf, false);           // row_newbcast:0
        const uint32_t x0h = __builtin_amdgcn_update_dpp(0u, (uint32_t)(x7 >> 32), 0x150, 0xf, 0xf, false);
        al += (uint64_t)x0l * c_elem0;
        ah += (uint64_t)x0h * c_elem0;
        GL_WAVE_LDS_SYNC();
        s = psd_recombine(al, ah);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) dense_round(r, true);
#endif
    return s;
}

// one 16-lane group = one parent node of layer `layer`; block = 64 threads = 4 nodes
template <int FORM>
__global__ void __launch_bounds__(64) merkle_level_lanes_kernel(uint64_t* digests, uint64_t* cap, uint32_t sub_bits,
                                                               uint32_t layer, uint64_t n_nodes) {
    __shared__ uint64_t rings[4][24];
    const int li = threadIdx.x & 15, grp = threadIdx.x >> 4;
    uint64_t g = blockIdx.x * 4ull + grp;
    const bool valid = g < n_nodes;
    if (!valid) g = 0;  // keep the whole wave in the barriers
    const uint64_t sub_leaves = 1ull << sub_bits;
    const uint64_t per = sub_leaves >> layer;
    const uint64_t t = g / per, k = g % per;
    uint64_t* tree = digests + t * 2 * (sub_leaves - 1) * 4;
    const uint64_t child = digest_slot(layer - 1, 2 * k);  // left child; the right one follows it
    uint64_t s = (li < 8) ? tree[child * 4 + li] : 0;
    s = psd_permute_lanes<FORM>(s, li, rings[grp]);
    if (valid && li < 4) {
        uint64_t* dst = (layer == sub_bits) ? cap + t * 4 : tree + digest_slot(layer, k) * 4;
        dst[li] = gl_canon(s);
    }
}

// The top of every cap subtree in ONE launch: one 1024-thread block per subtree takes its 2^(n_levels - 1) nodes of layer `layer0` (at most
// 128) and climbs n_levels levels to the cap entry, the digests of a level staying in LDS for the next one (and going to the plonky2
// layout in HBM).  Each of those levels is one lane-parallel permutation deep (two for the 128-node level: 64 groups of 16 lanes); as
// separate launches they cost ~20 us apiece, mostly launch latency -- round 2 ran the last six levels here and two more as separate
// lane-parallel launches (12.7 % of the kernel time of the 8-context profile between them); with eight levels a tree is leaf hashing,
// its wide levels, and this.  Trees of at most 2^8 leaves per cap entry are built whole.  Waves whose groups have no node on a level skip it.
#define MERKLE_TOP_LEVELS 8
template <int FORM>
__global__ void __launch_bounds__(1024) merkle_top_kernel(uint64_t* digests, uint64_t* cap, uint32_t sub_bits, uint32_t layer0, uint32_t n_levels) {
    __shared__ uint64_t rings[64][24];
    __shared__ uint64_t lvl[2][(1 << (MERKLE_TOP_LEVELS - 1)) * 4];
    const int li = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int wave_first_grp = (threadIdx.x >> 6) << 2;
    const uint64_t sub_leaves = 1ull << sub_bits;
    const uint64_t t = blockIdx.x;
    uint64_t* tree = digests + t * 2 * (sub_leaves - 1) * 4;
    int cur = 0;
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t layer = layer0 + l;
        const int cnt = 1 << (n_levels - 1 - l);
        for (int base = 0; base < cnt; base += 64) {
            if (base + wave_first_grp < cnt) {                // wave-uniform: this wave owns at least one node of the round
                const bool valid = base + grp < cnt;
                const int j = valid ? base + grp : 0;
                uint64_t s = 0;
                if (li < 8) s = (l == 0) ? tree[digest_slot(layer - 1, 2 * (uint64_t)j) * 4 + li] : lvl[cur][(2 * j) * 4 + li];
                s = psd_permute_lanes<FORM>(s, li, rings[grp]);
                if (valid && li < 4) {
                    const uint64_t v = gl_canon(s);
                    lvl[cur ^ 1][j * 4 + li] = v;
                    uint64_t* dst = (layer == sub_bits) ? cap + t * 4 : tree + digest_slot(layer, (uint64_t)j) * 4;
                    dst[li] = v;
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

__global__ void __launch_bounds__(256) two_to_one_kernel(const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t* out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s[12] = {l[4 * i], l[4 * i + 1], l[4 * i + 2], l[4 * i + 3],
                      r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], 0, 0, 0, 0};
    psd_permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) out[4 * i + k] = gl_canon(s[k]);
}

// proof of work: candidates start + g; the smallest passing candidate of the launch wins (atomicMin)
__global__ void __launch_bounds__(256) pow_grind_kernel(const uint64_t* state, uint32_t pos, uint32_t bits,
                                                       uint64_t start, unsigned long long* best) {
    const uint64_t w = start + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (w > __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // cannot be the smallest witness any more
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = state[k];
#pragma unroll
    for (int k = 0; k < 12; k++) if ((uint32_t)k == pos) s[k] = w;
    psd_permute(s);
    const uint64_t resp = gl_canon(s[7]);
    if (bits == 0 || (resp >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)w);
}

// query openings: block q copies row idx[q] of a column-major matrix and its Merkle path (siblings
// leaf -> cap, MerkleTree::prove's index walk) into dense output buffers -- one launch + one D2H per
// oracle for all FRI queries instead of (1 + depth) tiny copies per query.
__global__ void open_batch_kernel(const uint64_t* lde, uint64_t stride, uint32_t leaf_len, const uint64_t* digests,
                                  uint32_t log_n, uint32_t cap_height, const uint64_t* idx, uint32_t idx_shift,
                                  uint64_t* leaves_out, uint64_t leaf_out_stride, uint64_t* sib_out, uint64_t sib_out_stride) {
    // stride == 0: leaves are row-major [n][leaf_len]; otherwise column-major with that column stride
    const uint64_t index = idx[blockIdx.x] >> idx_shift;
    const uint32_t layers = log_n - cap_height;
    for (uint32_t c = threadIdx.x; c < leaf_len; c += blockDim.x)
        leaves_out[(uint64_t)blockIdx.x * leaf_out_stride + c] = stride ? lde[(uint64_t)c * stride + index] : lde[index * leaf_len + c];
    const uint64_t sub_leaves = 1ull << layers;
    const uint64_t* tree = digests + (index >> layers) * 2 * (sub_leaves - 1) * 4;
    const uint64_t k0 = index & (sub_leaves - 1);
    for (uint32_t e = threadIdx.x; e < layers * 4; e += blockDim.x) {
        const uint32_t layer = e >> 2;
        const uint64_t k = (k0 >> layer) ^ 1;  // sibling of the node on the path at this layer
        sib_out[(uint64_t)blockIdx.x * sib_out_stride + layer * 4 + (e & 3)] = tree[digest_slot(layer, k) * 4 + (e & 3)];
    }
}

int32_t open_batch_dev(Ctx* ctx, const uint64_t* lde, uint64_t stride, uint32_t leaf_len, const uint64_t* digests,
                       uint32_t log_n, uint32_t cap_height, const uint64_t* idx_dev, uint32_t n_idx, uint64_t* leaves_out,
                       uint64_t* sib_out) {
    if (n_idx == 0) return GL355_OK;
    ProfScope ps(ctx, "open_batch", (uint64_t)n_idx * (leaf_len * 16 + (log_n - cap_height) * 64));
    hipLaunchKernelGGL(open_batch_kernel, dim3(n_idx), dim3(64), 0, ctx->stream, lde, stride, leaf_len, digests, log_n,
                       cap_height, idx_dev, 0u, leaves_out, (uint64_t)leaf_len, sib_out, (uint64_t)(log_n - cap_height) * 4);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
// general form: index = idx[q] >> idx_shift, explicit output strides (FRI layer trees share one output row per query)
int32_t open_batch_ex_dev(Ctx* ctx, const uint64_t* leaves, uint64_t stride, uint32_t leaf_len, const uint64_t* digests,
                          uint32_t log_n, uint32_t cap_height, const uint64_t* idx_dev, uint32_t idx_shift, uint32_t n_idx,
                          uint64_t* leaves_out, uint64_t leaf_out_stride, uint64_t* sib_out, uint64_t sib_out_stride) {
    if (n_idx == 0) return GL355_OK;
    ProfScope ps(ctx, "open_batch", (uint64_t)n_idx * (leaf_len * 16 + (log_n - cap_height) * 64));
    hipLaunchKernelGGL(open_batch_kernel, dim3(n_idx), dim3(64), 0, ctx->stream, leaves, stride, leaf_len, digests, log_n,
                       cap_height, idx_dev, idx_shift, leaves_out, leaf_out_stride, sib_out, sib_out_stride);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ------------------------------------------------------------------------------------------------
int32_t poseidon_permute_dev(Ctx* ctx, uint64_t* states, uint64_t count) {
    if (count == 0) return GL355_OK;
    hipLaunchKernelGGL(poseidon_permute_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, states, count);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

static int32_t launch_leaves(Ctx* ctx, const LeafArgs& a) {
    if (a.n_leaves == 0) return GL355_OK;
    hipLaunchKernelGGL(hash_leaves_kernel, dim3((uint32_t)((a.n_leaves + 255) / 256)), dim3(256), 0, ctx->stream, a);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

int32_t hash_leaves_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major,
                        uint64_t col_stride, uint64_t* digests) {
    LeafArgs a;
    memset(&a, 0, sizeof a);
    a.leaves = leaves; a.n_leaves = n_leaves; a.leaf_len = leaf_len; a.col_major = col_major;
    a.stride = col_major ? col_stride : leaf_len;
    a.out = digests; a.linear = 1;
    return launch_leaves(ctx, a);
}

int32_t hash_no_pad_dev(Ctx* ctx, const uint64_t* inputs, uint64_t n, uint32_t len, uint64_t* digests) {
    LeafArgs a;
    memset(&a, 0, sizeof a);
    a.leaves = inputs; a.n_leaves = n; a.leaf_len = len; a.col_major = 0; a.stride = len;
    a.out = digests; a.linear = 1; a.always_hash = 1;
    return launch_leaves(ctx, a);
}

int32_t two_to_one_dev(Ctx* ctx, const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t* out) {
    if (n == 0) return GL355_OK;
    hipLaunchKernelGGL(two_to_one_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, l, r, n, out);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

int32_t merkle_build_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major,
                         uint64_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap) {
    const uint32_t log_n = log2_u64(n_leaves);
    if ((1ull << log_n) != n_leaves) return ctx->fail(GL355_E_INVALID_ARG, "merkle: n_leaves must be a power of two");
    if (cap_height > log_n) return ctx->fail(GL355_E_INVALID_ARG, "merkle: cap_height > log2(n_leaves)");
    LeafArgs a;
    memset(&a, 0, sizeof a);
    a.leaves = leaves; a.n_leaves = n_leaves; a.leaf_len = leaf_len; a.col_major = col_major;
    a.stride = col_major ? col_stride : leaf_len;
    return merkle_build_args(ctx, a, log_n - cap_height, digests, cap);
}

// a.n_leaves leaves (any multiple of 2^sub_bits: the cap subtrees of one tree, or of several units' trees laid out one after
// the other) -> per-subtree digests in plonky2's layout + one cap entry per subtree.  `a` describes where the leaves are.
int32_t merkle_build_args(Ctx* ctx, LeafArgs a, uint32_t sub_bits, uint64_t* digests, uint64_t* cap) {
    const uint64_t n_leaves = a.n_leaves;
    if (n_leaves == 0 || (n_leaves & ((1ull << sub_bits) - 1))) return ctx->fail(GL355_E_INVALID_ARG, "merkle: leaves do not fill whole cap subtrees");
    a.out = digests; a.cap = cap; a.sub_bits = sub_bits; a.linear = 0;
    { ProfScope ps(ctx, "hash_leaves_kernel", n_leaves * ((uint64_t)a.leaf_len * 8 + 32)); GL355_TRY(launch_leaves(ctx, a)); }
    const uint64_t lanes_max = 1ull << ctx->merkle_lanes_log;
    // the last MERKLE_TOP_LEVELS levels of every cap subtree go to merkle_top_kernel (when they are lane-parallel levels anyway: at most
    // lanes_max nodes enter it over the whole forest)
    // Eight levels per block cost three dependent permutations more than six levels behind two forest-wide lane-parallel launches: with
    // few subtrees (one proof: 16) the separate launches spread over the whole chip and the tree is done sooner (single-unit latency 17.0 vs
    // 17.6 ms); with a lock-step batch (>= 64 subtrees) the chip is full either way and four launches fewer per unit win
    // (profiles/r03_merkle_top_ab.txt).
    const uint32_t top_levels_max = (n_leaves >> sub_bits) < 64 ? 6 : MERKLE_TOP_LEVELS;
    // a lone proof's forest (16 cap subtrees) is latency-bound: the partial rounds with the row inside the S-box (psd_permute_lanes<3>); a lock-step batch
    // shares the chip with other contexts' hash kernels, where the 16 extra registers of that form cost more than its shorter chain gains
    const bool fast_lanes = PSD_LANES_FORM < 0 ? (n_leaves >> sub_bits) < 64 : PSD_LANES_FORM == 3;
    uint32_t top_levels = std::min<uint32_t>(top_levels_max, sub_bits);
    while (top_levels > 1 && ((n_leaves >> (sub_bits - top_levels + 1)) > lanes_max)) top_levels--;      // a forest of many subtrees: fewer levels each
    uint32_t top_from = sub_bits + 1;
    if (top_levels >= 1 && ((n_leaves >> (sub_bits - top_levels + 1)) <= lanes_max)) top_from = sub_bits - top_levels + 1;
    for (uint32_t layer = 1; layer <= sub_bits; layer++) {
        const uint64_t n_nodes = n_leaves >> layer;
        if (layer == top_from) {
            ProfScope ps(ctx, "merkle_top_kernel", (2 * n_nodes - (n_leaves >> sub_bits)) * 96);
            if (fast_lanes) hipLaunchKernelGGL(merkle_top_kernel<3>, dim3((uint32_t)(n_leaves >> sub_bits)), dim3(1024), 0, ctx->stream, digests, cap, sub_bits, layer, top_levels);
            else hipLaunchKernelGGL(merkle_top_kernel<0>, dim3((uint32_t)(n_leaves >> sub_bits)), dim3(1024), 0, ctx->stream, digests, cap, sub_bits, layer, top_levels);
            GL355_HIP(ctx, hipGetLastError());
            break;
        }
        // one scope per level (= per launch): 2 child digests in, 1 out per node
        ProfScope ps(ctx, n_nodes <= lanes_max ? "merkle_level_lanes_kernel" : "merkle_level_kernel", n_nodes * 96);
        if (n_nodes <= lanes_max) {
            // small level: 16 lanes per node (latency ~10x lower than one lane per node)
            if (fast_lanes) hipLaunchKernelGGL(merkle_level_lanes_kernel<3>, dim3((uint32_t)((n_nodes + 3) / 4)), dim3(64), 0, ctx->stream,
                                               digests, cap, sub_bits, layer, n_nodes);
            else hipLaunchKernelGGL(merkle_level_lanes_kernel<0>, dim3((uint32_t)((n_nodes + 3) / 4)), dim3(64), 0, ctx->stream,
                                    digests, cap, sub_bits, layer, n_nodes);
        } else {
            hipLaunchKernelGGL(merkle_level_kernel, dim3((uint32_t)((n_nodes + 255) / 256)), dim3(256), 0, ctx->stream,
                               digests, cap, sub_bits, layer, n_nodes);
        }
        GL355_HIP(ctx, hipGetLastError());
    }
    return GL355_OK;
}

int32_t pow_grind_dev(Ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start,
                      uint64_t* witness_host) {
    if (pos >= 8) return ctx->fail(GL355_E_INVALID_ARG, "pow: witness position must be in the rate part");
    if (bits > 40) return ctx->fail(GL355_E_UNSUPPORTED, "pow: more than 40 bits of grinding refused");
    Scratch sc(ctx);
    GL355_TRY(sc.get(13 * sizeof(uint64_t)));
    uint64_t* d_state = sc.as<uint64_t>();
    unsigned long long* d_best = reinterpret_cast<unsigned long long*>(d_state + 12);
    uint64_t host[13];
    for (int i = 0; i < 12; i++) host[i] = gl_canon(state[i]);
    host[12] = ~0ull;
    GL355_HIP(ctx, hipMemcpyAsync(d_state, host, sizeof host, hipMemcpyHostToDevice, ctx->stream));
    // batches of 2^20 candidates until one launch contains a solution; within a launch the
    // minimum wins, and earlier launches found nothing, so the result is the global minimum.
    // a launch of 2^(bits+1) candidates contains a solution with probability 1 - e^-2; grow when unlucky
    uint64_t per_launch = 1ull << std::min<uint32_t>(std::max<uint32_t>(bits + 1, 12), 22);
    uint64_t base = start;
    for (;;) {
        ProfScope ps(ctx, "pow_grind", 0);   // pure compute: one permutation per candidate, no HBM traffic
        hipLaunchKernelGGL(pow_grind_kernel, dim3((uint32_t)(per_launch / 256)), dim3(256), 0, ctx->stream, d_state, pos,
                           bits, base, d_best);
        GL355_HIP(ctx, hipGetLastError());
        unsigned long long best;
        GL355_HIP(ctx, ctx->d2h(&best, d_best, sizeof best));
        GL355_HIP(ctx, ctx->wait());
        if (best != ~0ull) { *witness_host = best; return GL355_OK; }
        base += per_launch;
        if (per_launch < (1ull << 22)) per_launch <<= 1;
        if (base - start > (1ull << 44)) return ctx->fail(GL355_E_UNSUPPORTED, "pow: no witness found in 2^44 candidates");
    }
}

}  // namespace gl355
