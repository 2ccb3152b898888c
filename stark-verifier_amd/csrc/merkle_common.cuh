// Pieces shared by the Merkle kernels of both hash back-ends (merkle.hip: Poseidon-Goldilocks, merkle_bn254.hip: the
// reference's Bn254PoseidonHash): plonky2's digest layout and the leaf-hash launch arguments.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace gl355 {

// index of node k of layer `layer` (0 = leaf digests) inside one cap-subtree's digest buffer:
// pair p = k>>1 of layer i sits at pair slot (p << (i+1)) + 2^i - 1 (MerkleTree::prove's formula).
__host__ __device__ __forceinline__ uint64_t digest_slot(uint32_t layer, uint64_t k) {
    return 2 * (((k >> 1) << (layer + 1)) + (1ull << layer) - 1) + (k & 1);
}

struct LeafArgs {
    const uint64_t* leaves;
    uint64_t n_leaves;      // all units together
    uint32_t leaf_len;
    uint32_t col_major;
    uint64_t stride;        // col-major: elements between columns; row-major: elements between rows
    uint64_t* out;          // digest destination
    uint32_t sub_bits;      // log2(leaves per cap subtree); layout = subtree t at out + t*sub_dig*4
    uint32_t linear;        // 1: out[i*4..] (no layout)
    uint32_t always_hash;   // hash_no_pad semantics (no <=4 shortcut)
    uint64_t* cap;          // used when sub_bits == 0 (tree is all cap)
    // several units (independent trees of 2^unit_log leaves each, the lock-step proofs of one prover context) in one launch:
    // leaf g belongs to unit g >> unit_log; its columns [0, n_main) come from `leaves + unit * unit_stride`, the remaining ones
    // (the salt columns of a blinded oracle) from `salt + unit * salt_unit_stride`.  unit_log == 0 means one unit, no salt segment.
    uint32_t unit_log;
    uint32_t n_main;
    uint64_t unit_stride;
    const uint64_t* salt;
    uint64_t salt_unit_stride;
};

// element k of leaf g
__device__ __forceinline__ uint64_t leaf_elem(const LeafArgs& a, uint64_t g, uint32_t k) {
    if (a.unit_log == 0) return a.col_major ? a.leaves[(uint64_t)k * a.stride + g] : a.leaves[g * a.stride + k];
    const uint64_t u = g >> a.unit_log, i = g & ((1ull << a.unit_log) - 1);
    if (k < a.n_main) {
        const uint64_t* base = a.leaves + u * a.unit_stride;
        return a.col_major ? base[(uint64_t)k * a.stride + i] : base[i * a.stride + k];
    }
    return a.salt[u * a.salt_unit_stride + (uint64_t)(k - a.n_main) * a.stride + i];     // salt columns are always column-major
}

}  // namespace gl355
