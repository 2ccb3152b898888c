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
    uint64_t n_leaves;
    uint32_t leaf_len;
    uint32_t col_major;
    uint64_t stride;        // col-major: elements between columns; row-major: elements between rows
    uint64_t* out;          // digest destination
    uint32_t sub_bits;      // log2(leaves per cap subtree); layout = subtree t at out + t*sub_dig*4
    uint32_t linear;        // 1: out[i*4..] (no layout)
    uint32_t always_hash;   // hash_no_pad semantics (no <=4 shortcut)
    uint64_t* cap;          // used when sub_bits == 0 (tree is all cap)
};

}  // namespace gl355
