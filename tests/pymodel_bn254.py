"""Big-integer model of the reference's BN254-Poseidon hasher over Goldilocks elements (test infrastructure).
Follows src/plonky2_verifier/bn245_poseidon/native.rs:16-77 (permutation, encode / decode) and
plonky2_config.rs:38-75 (Bn254PoseidonPermutation::permute, hash_no_pad, two_to_one)."""
import os
import sys

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # BN254 scalar field
PG = (1 << 64) - (1 << 32) + 1
T, RF, RP = 5, 8, 60
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "tools"))
from poseidon_grain import bn254_t5  # noqa: E402  (Grain-LFSR generation of the published parameters)

RC, MDS = bn254_t5()


def permute_fr(state, transposed=False):
    """native.rs:45-62; transposed=True mixes with M^T (circomlib's convention), only used to pin the constants"""
    s = [x % R for x in state]
    k = 0
    for rnd in range(RF + RP):
        s = [(x + RC[k + i]) % R for i, x in enumerate(s)]
        k += T
        if rnd < RF // 2 or rnd >= RF // 2 + RP:
            s = [pow(x, 5, R) for x in s]
        else:
            s[0] = pow(s[0], 5, R)
        if transposed:
            s = [sum(MDS[j][i] * s[j] for j in range(T)) % R for i in range(T)]
        else:
            s = [sum(MDS[i][j] * s[j] for j in range(T)) % R for i in range(T)]
    return s


def encode_fe(x3):            # native.rs:64-69: x0 + x1 p + x2 p^2
    return (x3[0] % PG + (x3[1] % PG) * PG + (x3[2] % PG) * PG * PG) % R


def decode_fe(x):             # native.rs:71-77 + chip/native_chip/utils.rs:25-36: the three low base-p digits
    out = []
    for _ in range(3):
        x, r = divmod(x, PG)
        out.append(r)
    return out


def permute(state12):         # plonky2_config.rs:38-55
    enc = [encode_fe(state12[3 * i: 3 * i + 3]) for i in range(4)] + [0]
    st = permute_fr(enc)
    flat = [d for x in st for d in decode_fe(x)]
    return flat[:12]


def hash_no_pad(xs):          # plonky2 hash_n_to_hash_no_pad over this permutation: overwrite-mode sponge, rate 8
    st = [0] * 12
    for off in range(0, len(xs), 8):
        chunk = [x % PG for x in xs[off:off + 8]]
        st[:len(chunk)] = chunk
        st = permute(st)
    return st[:4]


def two_to_one(l, r):         # plonky2 compress: permute(l | r | 0^4)[0..4]
    return permute(list(l) + list(r) + [0] * 4)[:4]
