"""Poseidon parameter generation (the Grain-LFSR procedure of the Poseidon paper's reference script
`generate_parameters_grain`): round constants by rejection sampling of n-bit strings, MDS = Cauchy matrix
1 / (x_i + y_j) over the next 2t distinct field elements.  For (prime field, x^alpha S-box, n = 254, t = 5, R_F = 8, R_P = 60)
over the BN254 scalar field this reproduces the published (circomlib) parameters the reference pins at
src/plonky2_verifier/bn245_poseidon/constants.rs:5-379 -- `tools/gen_bn254_tables.py --check-reference` compares all
340 + 25 values.  (The script's MDS security checks are not re-implemented: the first Cauchy candidate is the published
matrix.)"""

BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _grain_bits(field, sbox, n, t, r_f, r_p):
    bits = []
    for v, w in ((field, 2), (sbox, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
        bits += [int(c) for c in bin(v)[2:].zfill(w)]
    bits += [1] * 30

    def update():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb
    for _ in range(160):
        update()
    while True:
        nb = update()
        while nb == 0:          # a 0 discards the following bit
            update()
            nb = update()
        yield update()


def _take(gen, n):
    v = 0
    for _ in range(n):
        v = (v << 1) | next(gen)
    return v


def poseidon_parameters(prime, n_bits, t, r_f, r_p):
    """(round constants [(r_f + r_p) * t], mds [t][t]) for an x^alpha Poseidon instance over F_prime"""
    gen = _grain_bits(1, 0, n_bits, t, r_f, r_p)
    rc = []
    while len(rc) < (r_f + r_p) * t:
        x = _take(gen, n_bits)
        if x < prime:
            rc.append(x)
    while True:
        pts = [_take(gen, n_bits) % prime for _ in range(2 * t)]
        if len(set(pts)) == 2 * t:
            break
    xs, ys = pts[:t], pts[t:]
    mds = [[pow((xs[i] + ys[j]) % prime, -1, prime) for j in range(t)] for i in range(t)]
    return rc, mds


def bn254_t5():
    return poseidon_parameters(BN254_R, 254, 5, 8, 60)
