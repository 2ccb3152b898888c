"""CPU: the oracle of the SNARK-finalisation kernels (bn256::Fr FFT, bn256::G1 MSM) against published constants
(halo2curves' ROOT_OF_UNITY, the EIP-196 2*G vector) and against the big-integer model tests/pymodel_bn254_curve.py."""
import random

import numpy as np

import pymodel_bn254_curve as pm
from oracle_lib import Bn254Curve


def test_published_constants():
    assert pm.ROOT == pm.HALO2CURVES_ROOT_OF_UNITY
    assert pow(pm.ROOT, 1 << 28, pm.R) == 1 and pow(pm.ROOT, 1 << 27, pm.R) != 1
    assert pm.add(pm.G, pm.G) == pm.EIP196_2G and pm.mul(pm.G, 2) == pm.EIP196_2G
    assert pm.mul(pm.G, pm.R) is None                                   # G has order r


def test_oracle_g1_against_model_and_kat(orc):
    cv = Bn254Curve(orc)
    assert cv.mul(pm.G, 2) == pm.EIP196_2G and cv.add(pm.G, pm.G) == pm.EIP196_2G
    assert cv.on_curve(pm.G) and cv.on_curve(pm.EIP196_2G) and not cv.on_curve((1, 3))
    rnd = random.Random(0x254)
    pts = [pm.mul(pm.G, rnd.randrange(1, pm.R)) for _ in range(6)]
    for p in pts:
        k = rnd.randrange(pm.R)
        assert cv.mul(p, k) == pm.mul(p, k)
    assert cv.add(pts[0], pts[1]) == pm.add(pts[0], pts[1])
    neg = (pts[2][0], pm.Q - pts[2][1])
    assert cv.add(pts[2], neg) is None and cv.add(None, pts[3]) == pts[3]
    assert cv.mul(pts[4], pm.R) is None and cv.mul(pts[4], 0) is None
    sc = [rnd.randrange(1 << 256) for _ in pts]                         # any 256-bit scalar is accepted
    assert cv.msm(pts + [None], sc + [5]) == pm.msm(pts, sc)
    # structured inputs for the large GPU tests: (first + i * step) G
    mult = cv.multiples(3, 5, 9)
    assert mult == [pm.mul(pm.G, 3 + 5 * i) for i in range(9)]


def test_oracle_fr_fft_against_definition(orc):
    cv = Bn254Curve(orc)
    rnd = random.Random(0x255)
    for log_n in (1, 2, 3, 5):
        a = [rnd.randrange(pm.R) for _ in range(1 << log_n)]
        assert cv.ntt(a) == pm.dft(a)
        assert cv.ntt(a, inverse=True) == pm.dft(a, inverse=True)
    a = [rnd.randrange(pm.R) for _ in range(1 << 10)]
    assert cv.ntt(cv.ntt(a), inverse=True) == a
    # a unit impulse at index 1 transforms to the powers of omega
    imp = [0, 1] + [0] * 14
    assert cv.ntt(imp) == [pow(pm.omega(4), k, pm.R) for k in range(16)]
