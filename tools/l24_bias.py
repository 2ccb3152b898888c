#!/usr/bin/env python3
"""The bias vector of csrc/ntt_l24.cuh (L24_BETA): a vector beta in Z^4 with sum_i beta_i 2^(24 i) = 0 (mod p) and every component close
to 1.5 * 2^28, so that limbs |l_i| < 2^28 become non-negative (and < 2^30) by adding it without changing the value mod p.
LLL on the lattice {v : sum v_i X^i = 0 mod p}, X = 2^24, then Babai's nearest plane to (1.5 * 2^28, ...).  Prints the vector and checks it."""
from fractions import Fraction as F

P = 2**64 - 2**32 + 1
X = 2**24


def dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def gram_schmidt(B):
    Bs = []
    for i in range(len(B)):
        v = [F(x) for x in B[i]]
        for j in range(i):
            m = dot(B[i], Bs[j]) / dot(Bs[j], Bs[j])
            v = [a - m * b for a, b in zip(v, Bs[j])]
        Bs.append(v)
    return Bs


def lll(B, delta=F(3, 4)):
    B = [list(r) for r in B]
    n, k = len(B), 1
    while k < n:
        for j in range(k - 1, -1, -1):
            Bs = gram_schmidt(B)
            q = round(dot(B[k], Bs[j]) / dot(Bs[j], Bs[j]))
            if q:
                B[k] = [a - q * b for a, b in zip(B[k], B[j])]
        Bs = gram_schmidt(B)
        mu = dot(B[k], Bs[k - 1]) / dot(Bs[k - 1], Bs[k - 1])
        if dot(Bs[k], Bs[k]) >= (delta - mu * mu) * dot(Bs[k - 1], Bs[k - 1]):
            k += 1
        else:
            B[k], B[k - 1] = B[k - 1], B[k]
            k = max(k - 1, 1)
    return B


def main():
    basis = [[P, 0, 0, 0], [-X, 1, 0, 0], [-(X * X) % P, 0, 1, 0], [-(X ** 3) % P, 0, 0, 1]]
    R = lll(basis)
    target = [3 * 2**27] * 4
    Bs = gram_schmidt(R)
    b = [F(x) for x in target]
    for i in range(3, -1, -1):
        c = round(dot(b, Bs[i]) / dot(Bs[i], Bs[i]))
        b = [a - c * x for a, x in zip(b, R[i])]
    beta = [int(t - r) for t, r in zip(target, b)]
    assert sum(v * X**i for i, v in enumerate(beta)) % P == 0
    assert all(2**28 <= v < 2**29 for v in beta)
    print("reduced basis:", R)
    print("L24_BETA =", beta)
    return beta


if __name__ == "__main__":
    assert main() == [402653208, 402653160, 402653160, 402653160]
