// Host-side bn256::Fr for the sequential parts of the Halo2 / KZG prover: challenges, the handful of scalars around every
// kernel launch, low-degree interpolation of the SHPLONK remainders, and the Keccak-256 transcript that halo2-solidity-verifier's
// Keccak256Transcript defines (chip/native_chip/test_utils.rs:73).  4 x 64-bit limbs, Montgomery form with R = 2^256 -- the same
// representation the kernels keep in HBM, so a value crosses the boundary as 32 bytes without conversion.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#ifndef BN254C_QUAL
#define BN254C_QUAL static const
#endif
#include "bn254_curve_tables.h"

namespace gl355 {

struct Fr {
    uint64_t l[4];          // Montgomery form, canonical (< r)
    typedef unsigned __int128 u128;
    static const uint64_t* M() { return BN254C_FR_MOD_64; }
    static bool geq_m(const uint64_t a[4]) {
        for (int i = 3; i >= 0; i--) { if (a[i] > M()[i]) return true; if (a[i] < M()[i]) return false; }
        return true;
    }
    static void sub_m(uint64_t a[4]) {
        u128 br = 0;
        for (int i = 0; i < 4; i++) { const u128 d = (u128)a[i] - M()[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
    }
    static Fr zero() { Fr r; memset(r.l, 0, 32); return r; }
    static Fr one() { Fr r; memcpy(r.l, BN254C_FR_ONE_64, 32); return r; }
    static Fr mont_mul(const Fr& a, const Fr& b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * BN254C_FR_N0INV_64;
            c = ((u128)m * M()[0] + t[0]) >> 64;
            for (int j = 1; j < 4; j++) { c += (u128)m * M()[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fr r = {{t[0], t[1], t[2], t[3]}};
        if (t[4] || geq_m(r.l)) sub_m(r.l);
        return r;
    }
    // any 256-bit integer (little-endian limbs) -> the field element
    static Fr from_words(const uint64_t w[4]) {
        Fr a; memcpy(a.l, w, 32);
        while (geq_m(a.l)) sub_m(a.l);
        Fr r2; memcpy(r2.l, BN254C_FR_R2_64, 32);
        return mont_mul(a, r2);
    }
    static Fr from_u64(uint64_t v) { const uint64_t w[4] = {v, 0, 0, 0}; return from_words(w); }
    // Montgomery words as the kernels store them (possibly lazily reduced, < 2r) -> canonical
    static Fr from_mont_words(const uint64_t w[4]) { Fr a; memcpy(a.l, w, 32); while (geq_m(a.l)) sub_m(a.l); return a; }
    void to_words(uint64_t w[4]) const { Fr o = {{1, 0, 0, 0}}; const Fr r = mont_mul(*this, o); memcpy(w, r.l, 32); }
    Fr operator*(const Fr& b) const { return mont_mul(*this, b); }
    Fr operator+(const Fr& b) const {
        Fr r; u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        if (c || geq_m(r.l)) sub_m(r.l);
        return r;
    }
    Fr operator-(const Fr& b) const {
        Fr r; u128 br = 0;
        for (int i = 0; i < 4; i++) { const u128 d = (u128)l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
        if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + M()[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    Fr neg() const { return zero() - *this; }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fr& b) const { return memcmp(l, b.l, 32) == 0; }
    bool operator!=(const Fr& b) const { return !(*this == b); }
    Fr pow(const uint64_t e[4]) const {
        Fr r = one();
        for (int i = 255; i >= 0; i--) {
            r = r * r;
            if ((e[i >> 6] >> (i & 63)) & 1) r = r * *this;
        }
        return r;
    }
    Fr pow_u64(uint64_t e) const { const uint64_t w[4] = {e, 0, 0, 0}; return pow(w); }
    Fr inv() const { const uint64_t e[4] = {M()[0] - 2, M()[1], M()[2], M()[3]}; return pow(e); }
    // canonical integer comparison (the order halo2curves' Ord gives Fr and BTreeSet iterates in)
    bool less_than(const Fr& b) const {
        uint64_t x[4], y[4];
        to_words(x); b.to_words(y);
        for (int i = 3; i >= 0; i--) { if (x[i] != y[i]) return x[i] < y[i]; }
        return false;
    }
    static Fr root_of_unity(uint32_t log_n) {
        Fr w = from_words(BN254C_FR_ROOT_64);
        for (uint32_t k = log_n; k < BN254C_FR_S; k++) w = w * w;
        return w;
    }
};

void keccak256_host(const uint8_t* data, size_t len, uint8_t out[32]);

// halo2-solidity-verifier's Keccak256Transcript (writer side): 32-byte big-endian words into a running buffer and into the proof; a
// challenge = keccak256(buffer) as a big-endian integer mod r, the digest becomes the buffer (plus 0x01 when squeezed twice in a row)
struct KeccakTranscript {
    std::vector<uint8_t> buf, proof;
    static void be32(const uint64_t w[4], uint8_t out[32]) {
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[31 - (8 * i + b)] = (uint8_t)(w[i] >> (8 * b));
    }
    void common_scalar(const Fr& s) { uint64_t w[4]; s.to_words(w); uint8_t b[32]; be32(w, b); buf.insert(buf.end(), b, b + 32); }
    void write_scalar(const Fr& s) { uint64_t w[4]; s.to_words(w); uint8_t b[32]; be32(w, b); buf.insert(buf.end(), b, b + 32); proof.insert(proof.end(), b, b + 32); }
    void write_point(const uint64_t xy[8]) {            // affine, canonical integers; all zero = the identity
        uint8_t b[64];
        be32(xy, b); be32(xy + 4, b + 32);
        buf.insert(buf.end(), b, b + 64);
        proof.insert(proof.end(), b, b + 64);
    }
    Fr squeeze_challenge() {
        std::vector<uint8_t> data(buf);
        if (buf.size() == 32) data.push_back(1);
        uint8_t h[32];
        keccak256_host(data.data(), data.size(), h);
        buf.assign(h, h + 32);
        // 256-bit big-endian integer mod r
        uint64_t w[4];
        for (int i = 0; i < 4; i++) { w[i] = 0; for (int b = 0; b < 8; b++) w[i] |= (uint64_t)h[31 - (8 * i + b)] << (8 * b); }
        return Fr::from_words(w);
    }
};

}  // namespace gl355
