"""CPU: the zero-knowledge blinding source.  The oracle's ChaCha20 block function is pinned by RFC 8439's own test vector
(section 2.3.2); the oracle's stream -> field-element convention is checked against Python integers; the product's host-side key
derivation (gl355_derive_key) equals the oracle's.  The device kernel is checked against the oracle in tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np

from oracle_lib import key_bytes

P = (1 << 64) - (1 << 32) + 1

# RFC 8439 2.3.2: key 00..1f, nonce 00:00:00:09 00:00:00:4a 00:00:00:00, block counter 1
RFC_KEY = bytes(range(32))
RFC_NONCE = (0x09000000, 0x4a000000, 0x00000000)
RFC_BLOCK = bytes.fromhex(
    "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
    "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def chacha_block(orc, key, counter, nonce):
    out = C.create_string_buffer(64)
    orc.L.orc_chacha20_block(key, C.c_uint32(counter), (C.c_uint32 * 3)(*nonce), out)
    return out.raw


def test_chacha20_block_rfc8439_vector(orc):
    assert chacha_block(orc, RFC_KEY, 1, RFC_NONCE) == RFC_BLOCK


def test_stream_elements_follow_the_documented_convention(orc):
    key = key_bytes(0xC0FFEE)
    out = np.zeros(37, dtype=np.uint64)
    orc.L.orc_blinding_elements(key, C.c_uint32(3), C.c_uint64(out.size), out.ctypes.data_as(C.c_void_p))
    for k in range(out.size):
        blk = chacha_block(orc, key, k // 4, (3, 0, 0))
        v = int.from_bytes(blk[16 * (k % 4):16 * (k % 4) + 16], "little")
        assert int(out[k]) == v % P
    # other stream, other key: unrelated values
    o2 = np.zeros(37, dtype=np.uint64)
    orc.L.orc_blinding_elements(key, C.c_uint32(4), C.c_uint64(o2.size), o2.ctypes.data_as(C.c_void_p))
    assert not np.array_equal(out, o2)


def test_derive_key_matches_the_oracle(gl, orc):
    lib = gl._lib.load()
    base = key_bytes(12345)
    for index in (0, 1, 2, 0xFFFFFFFF, 0x1_0000_0007):
        a, b = C.create_string_buffer(32), C.create_string_buffer(32)
        assert lib.gl355_derive_key(base, C.c_uint64(index), a) == 0
        orc.L.orc_derive_key(base, C.c_uint64(index), b)
        assert a.raw == b.raw
        assert a.raw == chacha_block(orc, base, 0, (0x0079656B, index & 0xFFFFFFFF, index >> 32))[:32]
    assert lib.gl355_derive_key(None, C.c_uint64(0), a) == -1
