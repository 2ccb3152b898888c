// Zero-knowledge blinding source: ChaCha20 in counter mode under a 256-bit per-proof key.
//
// The reference proves with `zero_knowledge: true` (src/plonky2_semaphore/access_set.rs:69, recursion.rs:33): plonky2 draws
// the salt columns of every blinded oracle (SALT_SIZE = 4, src/plonky2_verifier/types/assigned.rs:67-71) and the blinding
// rows of the witness from an OS-seeded CSPRNG.  Salt values are PUBLISHED in every FRI query leaf, so they must not be
// invertible to their seed and must not share a stream with the wire blinding.  Here:
//   * one 256-bit key per proof (caller-supplied, or drawn from getrandom(2) when the caller passes NULL);
//   * ChaCha20 block function of RFC 8439 2.3 (32-bit block counter, 96-bit nonce), nonce = (stream id, 0, 0);
//   * streams: 1 = wires salt, 2 = Z / partial-products salt, 3 = quotient salt, 4 = witness blinding rows;
//   * element k of a stream = the 16 key-stream bytes [16k, 16k + 16) read as little-endian lo | hi << 64, reduced mod p
//     (bias 2^-64); four elements per 64-byte block;
//   * per-unit keys of the batch runtime: key_j = first 32 bytes of block 0 under the batch key with nonce = ("key", j_lo, j_hi).
// oracle/gl_prover.c restates the same convention independently, so (witness, key) fixes the proof bytes on both sides.
#pragma once
#include <stdint.h>

#include "gl_field.cuh"

namespace gl355 {

#define GL355_BLIND_STREAM_WIRES_SALT 1u
#define GL355_BLIND_STREAM_ZS_SALT 2u
#define GL355_BLIND_STREAM_QUOTIENT_SALT 3u
#define GL355_BLIND_STREAM_WITNESS 4u
#define GL355_BLIND_NONCE_KEY 0x0079656bu   // "key\0": per-unit key derivation

struct BlindKey { uint32_t w[8]; };

GL_HD uint32_t chacha_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define GL355_CHACHA_QR(a, b, c, d)            \
    a += b; d ^= a; d = chacha_rotl(d, 16);    \
    c += d; b ^= c; b = chacha_rotl(b, 12);    \
    a += b; d ^= a; d = chacha_rotl(d, 8);     \
    c += d; b ^= c; b = chacha_rotl(b, 7);

// RFC 8439 2.3: the 16 output words of one block
GL_HD void chacha20_block(const BlindKey& key, uint32_t counter, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t out[16]) {
    const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.w[0], key.w[1], key.w[2], key.w[3],
                             key.w[4], key.w[5], key.w[6], key.w[7], counter, n0, n1, n2};
    uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
    uint32_t x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
    for (int i = 0; i < 10; i++) {
        GL355_CHACHA_QR(x0, x4, x8, x12) GL355_CHACHA_QR(x1, x5, x9, x13) GL355_CHACHA_QR(x2, x6, x10, x14) GL355_CHACHA_QR(x3, x7, x11, x15)
        GL355_CHACHA_QR(x0, x5, x10, x15) GL355_CHACHA_QR(x1, x6, x11, x12) GL355_CHACHA_QR(x2, x7, x8, x13) GL355_CHACHA_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + in[0]; out[1] = x1 + in[1]; out[2] = x2 + in[2]; out[3] = x3 + in[3];
    out[4] = x4 + in[4]; out[5] = x5 + in[5]; out[6] = x6 + in[6]; out[7] = x7 + in[7];
    out[8] = x8 + in[8]; out[9] = x9 + in[9]; out[10] = x10 + in[10]; out[11] = x11 + in[11];
    out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}

// the four field elements of block `block` of stream `stream`
GL_HD void blind_block_elements(const BlindKey& key, uint32_t stream, uint32_t block, uint64_t e[4]) {
    uint32_t o[16];
    chacha20_block(key, block, stream, 0, 0, o);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; j++) {
        const uint64_t lo = (uint64_t)o[4 * j] | ((uint64_t)o[4 * j + 1] << 32), hi = (uint64_t)o[4 * j + 2] | ((uint64_t)o[4 * j + 3] << 32);
        e[j] = gl_canon(gl_reduce128(lo, hi));
    }
}

static inline BlindKey blind_key_from_bytes(const uint8_t b[32]) {
    BlindKey k;
    for (int i = 0; i < 8; i++) k.w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return k;
}

}  // namespace gl355
