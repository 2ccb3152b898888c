// Keccak-256 on the host: the hash behind halo2-solidity-verifier's Keccak256Transcript, which the reference's finalisation
// proof is written through (chip/native_chip/test_utils.rs:73, verifier_api.rs:90) so that the EVM verifier can replay it with the KECCAK256
// opcode.  Original Keccak padding (0x01 ... 0x80), rate 136 bytes -- not NIST SHA3-256.  A few hundred bytes per challenge: host work.
#include <stdint.h>
#include <string.h>

#include "../../include/gl355.h"
#include "host_fr.h"

namespace gl355 {
namespace {
const uint64_t KRC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                          0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                          0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                          0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                          0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
inline uint64_t rotl(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }
void keccak_f(uint64_t st[25]) {
    for (int r = 0; r < 24; r++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            const uint64_t t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            const int j = KPIL[i];
            const uint64_t b = st[j];
            st[j] = rotl(t, KROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KRC[r];
    }
}
}  // namespace

void keccak256_host(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint64_t st[25];
    memset(st, 0, sizeof st);
    size_t off = 0;
    uint8_t block[136];
    for (;;) {
        const size_t take = len - off < rate ? len - off : rate;
        memset(block, 0, rate);
        if (take) memcpy(block, data + off, take);
        const bool last = take < rate;
        if (last) { block[take] ^= 0x01; block[rate - 1] ^= 0x80; }
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w = 0;
            for (int b = 0; b < 8; b++) w |= (uint64_t)block[8 * i + b] << (8 * b);
            st[i] ^= w;
        }
        keccak_f(st);
        off += take;
        if (last) break;
    }
    for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(st[i] >> (8 * b));
}

}  // namespace gl355

extern "C" int32_t gl355_keccak256(const uint8_t* data, uint64_t len, uint8_t out[32]) {
    if ((!data && len) || !out) return GL355_E_INVALID_ARG;
    gl355::keccak256_host(data, (size_t)len, out);
    return GL355_OK;
}
