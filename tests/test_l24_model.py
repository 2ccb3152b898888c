"""CPU: the algebra behind csrc/ntt_l24.cuh (the LDE column pass on carry-free 24-bit limbs), restated in Python integers and checked against
the definitions the reference's verifier fixes (omega_N = 7^((p-1)/N), chip/fri_chip.rs:162-163): the bias vector, the limb rotations and signs
of the radix-8 network, the shift twiddles between the two radix-8 rounds of a 32- / 64-point transform, the exit sum, and the magnitude bounds
that keep every limb inside int32 and every multiply-add chain inside 64 bits.  (The kernels themselves are checked bit for bit on the GPU:
tests/test_gpu_parity.py / test_gpu_large.py, every LDE.)"""
import random

P = 2**64 - 2**32 + 1
X = 1 << 24
BETA = [402653208, 402653160, 402653160, 402653160]          # csrc/ntt_l24.cuh L24_BETA; derivation: tools/l24_bias.py


def value(l):
    return sum(v * X**i for i, v in enumerate(l)) % P


def split(x):
    return [x & 0xFFFFFF, (x >> 24) & 0xFFFFFF, x >> 48, 0]


def bfly(a, b, rho, neg):
    """l24_bfly<RHO, NEG>: (a + b, +-(a - b) * X^rho) with the rotation folded into the operand order"""
    s = [a[i] + b[i] for i in range(4)]
    d = []
    for i in range(4):
        j = (i - rho) & 3
        wrap = i < rho
        d.append(b[j] - a[j] if (neg != wrap) else a[j] - b[j])
    return s, d


FWD = {0: (0, False), 2: (1, True), 4: (2, False), 6: (3, True)}       # omega_16^E as (limb rotation, sign): l24_bfly_w


def dif8(y):
    y = [list(v) for v in y]
    for pairs in ([(0, 4, 0), (1, 5, 2), (2, 6, 4), (3, 7, 6)], [(0, 2, 0), (1, 3, 4), (4, 6, 0), (5, 7, 4)], [(0, 1, 0), (2, 3, 0), (4, 5, 0), (6, 7, 0)]):
        for i, j, e in pairs:
            y[i], y[j] = bfly(y[i], y[j], *FWD[e])
    return y


def dif4(y):
    y = [list(v) for v in y]
    for pairs in ([(0, 2, 0), (1, 3, 4)], [(0, 1, 0), (2, 3, 0)]):
        for i, j, e in pairs:
            y[i], y[j] = bfly(y[i], y[j], *FWD[e])
    return y


def shift(e, s):
    """l24_shift<S>"""
    sg, a, b = s >= 96, (s % 96) // 24, (s % 96) % 24
    g = [v & ((1 << (24 - b)) - 1) for v in e]
    h = [v >> (24 - b) for v in e]
    y = [(g[0] << b) - h[3]] + [(g[i] << b) + h[i - 1] for i in range(1, 4)]
    r = [0] * 4
    for i in range(4):
        r[(i + a) & 3] = -y[i] if (sg != (i + a >= 4)) else y[i]
    return r


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def test_bias_vector():
    assert value(BETA) == 0 and all(2**28 <= v < 2**29 for v in BETA)
    # four products (l + beta) * 32-bit word cannot overflow a 64-bit accumulator for |l| < 2^28
    assert 4 * (2**28 + max(BETA)) * (2**32 - 1) < 2**64
    assert min(BETA) - 2**28 >= 0


def test_roots_are_powers_of_two():
    for k, e in ((1, 96), (2, 48), (3, 120), (4, 156), (5, 78), (6, 39)):
        assert pow(7, (P - 1) >> k, P) == pow(2, e, P)
    assert pow(2, 96, P) == P - 1


def test_split_shift_value():
    rnd = random.Random(1)
    for _ in range(200):
        x = rnd.randrange(1 << 64)
        l = split(x)
        assert value(l) == x % P
        big = [rnd.randrange(-(1 << 27), 1 << 27) for _ in range(4)]
        for s in (0, 3, 39, 78, 96, 117, 156, 189):
            r = shift(big, s)
            assert value(r) == value(big) * pow(2, s, P) % P
            assert all(abs(v) < (1 << 25) for v in r), (s, r)                    # renormalised: |out| < 2^24 + 2^27 2^(b-24) <= 2^25


def test_radix8_network_is_the_dft():
    rnd = random.Random(2)
    w8 = pow(7, (P - 1) >> 3, P)
    x = [rnd.randrange(P) for _ in range(8)]
    y = dif8([split(v) for v in x])
    for pos in range(8):
        k = brev(pos, 3)
        assert value(y[pos]) == sum(x[j] * pow(w8, j * k, P) for j in range(8)) % P
        assert all(abs(v) <= 8 * (1 << 24) for v in y[pos])
    w4 = pow(7, (P - 1) >> 2, P)
    z = dif4([split(v) for v in x[:4]])
    for pos in range(4):
        assert value(z[pos]) == sum(x[j] * pow(w4, j * brev(pos, 2), P) for j in range(4)) % P


def super_round(x, m):
    """2^m points (m = 6: 8 x 8, m = 5: 8 x 4) the way the kernels run them: radix-8 over q for every r, the shift twiddles 2^(mult r k0),
    then radix-8 / radix-4 over r; returns limb values in DIF (bit-reversed) order"""
    mult = 39 if m == 6 else 78
    nr = 1 << (m - 3)
    tile = [None] * (1 << m)
    for r in range(nr):
        y = dif8([split(x[nr * q + r]) for q in range(8)])
        for q in range(8):
            tile[nr * q + r] = shift(y[q], (mult * r * brev(q, 3)) % 192)
    out = [None] * (1 << m)
    for q in range(8):
        z = (dif8 if m == 6 else dif4)([tile[nr * q + r] for r in range(nr)])
        for s_ in range(nr):
            out[nr * q + s_] = z[s_]
    return out


def test_super_rounds_are_the_dft_and_stay_in_range():
    rnd = random.Random(3)
    for m in (5, 6):
        n = 1 << m
        w = pow(7, (P - 1) >> m, P)
        for trial in range(3):
            x = [P - 1] * n if trial == 0 else [rnd.randrange(P) for _ in range(n)]
            out = super_round(x, m)
            for pos in range(0, n, 5):
                k = brev(pos, m)
                assert value(out[pos]) == sum(x[j] * pow(w, j * k, P) for j in range(n)) % P
            assert all(abs(v) < (1 << 28) for l in out for v in l)              # the bound the exit sum relies on
            # the exit: sum (l + beta) 2^(24 i) with non-negative operands
            for l in out[:8]:
                assert all(0 <= v + b < (1 << 30) for v, b in zip(l, BETA))
                assert sum((v + b) * X**i for i, (v, b) in enumerate(zip(l, BETA))) % P == value(l)
