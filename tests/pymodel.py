"""Independent big-integer model of the hot path's mathematics (pure Python, small sizes only).

It shares no code with oracle/ or the HIP library: transforms are evaluated from their DEFINITION
(out[j] = sum_i c_i w^(ij), P(shift * w^j), recursive Merkle, Horner), so agreement with it pins the
oracle "by definition" where the reference holds no golden vectors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon_tables as gpt  # noqa: E402  (naive Poseidon from the published constants)

P = (1 << 64) - (1 << 32) + 1
_RC = gpt.load_rc()


def root_of_unity(log_n):          # chip/fri_chip.rs:162-163
    return pow(7, (P - 1) >> log_n, P)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def dft(c, inverse=False):
    n = len(c)
    lg = n.bit_length() - 1
    w = root_of_unity(lg)
    if inverse:
        w = pow(w, P - 2, P)
    out = [sum(c[i] * pow(w, i * j, P) for i in range(n)) % P for j in range(n)]
    if inverse:
        ninv = pow(n, P - 2, P)
        out = [x * ninv % P for x in out]
    return out


def poly_eval(c, x):
    acc = 0
    for a in reversed(c):
        acc = (acc * x + a) % P
    return acc


def lde(c, rate_bits, shift=7):    # out[j] = P(shift * w_N^j)
    n = len(c)
    N = n << rate_bits
    w = root_of_unity(N.bit_length() - 1)
    return [poly_eval(c, shift * pow(w, j, P) % P) for j in range(N)]


def permute(state):
    return gpt.permute_naive(list(state), _RC)


def hash_no_pad(x):                # chip/hasher_chip.rs:122-148
    st = [0] * 12
    for off in range(0, len(x), 8):
        chunk = x[off:off + 8]
        st[:len(chunk)] = [v % P for v in chunk]
        st = permute(st)
    return st[:4]


def hash_or_noop(x):               # chip/merkle_proof_chip.rs:52-57
    return ([v % P for v in x] + [0] * 4)[:4] if len(x) <= 4 else hash_no_pad(x)


def two_to_one(l, r):
    return permute(list(l) + list(r) + [0] * 4)[:4]


def merkle(leaves, cap_height):
    """returns (digests in plonky2's recursive layout, cap)."""
    n = len(leaves)
    sub = n >> cap_height

    def fill(lv):
        if len(lv) == 1:
            return [], hash_or_noop(lv[0])
        dl, l = fill(lv[:len(lv) // 2])
        dr, r = fill(lv[len(lv) // 2:])
        return dl + [l, r] + dr, two_to_one(l, r)
    digests, cap = [], []
    for t in range(1 << cap_height):
        d, root = fill(leaves[t * sub:(t + 1) * sub])
        digests += d
        cap.append(root)
    return digests, cap


# ---- quadratic extension, X^2 = 7 ----------------------------------------------------------------
def ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def ext_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def ext_inv(a):
    norm = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    ni = pow(norm, P - 2, P)
    return (a[0] * ni % P, (-a[1]) * ni % P)


def ext_poly_eval(c, x):           # c: list of ext pairs (or ints), x ext
    acc = (0, 0)
    for a in reversed(c):
        a = a if isinstance(a, tuple) else (a % P, 0)
        acc = ext_add(ext_mul(acc, x), a)
    return acc
