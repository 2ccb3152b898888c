"""Big-integer model of the BN254 (alt_bn128 / halo2curves bn256) pieces behind SURVEY 8(f) N4: Fr FFT by its definition and G1
arithmetic in affine coordinates with modular inverses -- nothing shared with oracle/bn254_curve_oracle.c or the HIP kernels."""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
S = 28
ROOT = pow(7, (R - 1) >> S, R)
# published constants the derivations must reproduce (tests/test_bn254_curve_oracle.py)
HALO2CURVES_ROOT_OF_UNITY = 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
EIP196_2G = (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
             0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
G = (1, 2)


def omega(log_n, inverse=False):
    w = pow(ROOT, 1 << (S - log_n), R)
    return pow(w, -1, R) if inverse else w


def dft(a, inverse=False):
    """a[k] = sum_i a[i] w^(ik) straight from the definition (small n only)"""
    n = len(a)
    log_n = n.bit_length() - 1
    w = omega(log_n, inverse)
    out = [sum(a[i] * pow(w, i * k, R) for i in range(n)) % R for k in range(n)]
    if inverse:
        ninv = pow(n, -1, R)
        out = [x * ninv % R for x in out]
    return out


def add(p, q):
    """affine addition; None is the identity"""
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return x3, (lam * (x1 - x3) - y1) % Q


def mul(p, k):
    r = None
    while k:
        if k & 1:
            r = add(r, p)
        p = add(p, p)
        k >>= 1
    return r


def msm(points, scalars):
    acc = None
    for p, k in zip(points, scalars):
        acc = add(acc, mul(p, k % R))
    return acc


def limbs4(v):
    return [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]


def from_limbs(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))
