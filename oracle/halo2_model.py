"""TEST INFRASTRUCTURE (the checker, never the product): a CPU restatement of halo2_proofs' `create_proof::<KZGCommitmentScheme<Bn256>,
ProverSHPLONK<_>, _, _, Keccak256Transcript, _>` -- the call the reference's SNARK finalisation makes
(/root/reference/src/plonky2_verifier/chip/native_chip/test_utils.rs:57-95, verifier_api.rs:77-92; README.md:171-177: 505-511 s at k = 23)
-- over Python integers, stage by stage, with the two heavy primitives (Fr FFT, G1 MSM) taken from oracle/bn254_curve_oracle.c.

PARITY UNPINNED: halo2_proofs, halo2curves and halo2-solidity-verifier are un-vendored git dependencies of the reference
(/root/reference/Cargo.lock: halo2_proofs from privacy-scaling-explorations/halo2 v2023_04_20, halo2curves 0.3.x, halo2-solidity-verifier),
their sources are absent from /root/reference and no Rust toolchain exists here, and the reference holds no proof fixture.  This file restates
the PUBLISHED algorithm of that halo2 version (plonk/prover.rs, plonk/{permutation,lookup,vanishing}/prover.rs, plonk/evaluation.rs,
poly/domain.rs, poly/kzg/multiopen/shplonk/prover.rs, and halo2-solidity-verifier's Keccak256Transcript), stage names and orders as in
those files; what pins it here is algebra: tests/halo2_verifier.py restates `verify_proof` + `VerifierSHPLONK` independently and accepts its
proofs (pairing checks evaluated in the exponent under a known tau), and rejects tampered ones.

Conventions that a byte-for-byte comparison with upstream would need and that cannot be checked here are marked [RECALLED].
Randomness: halo2 draws blinding values from the caller's RngCore; here every random scalar is a function of (seed, stream, a, index)
through ChaCha20 (RFC 8439 block function; 64 key-stream bytes -> 512-bit little-endian integer mod r, the map of Fr::from_uniform_bytes),
the same convention the product documents in include/gl355.h, so (witness, seed) fixes the proof bytes on both sides.
"""
import ctypes as C
import os
import struct

import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
S = 28
ROOT_OF_UNITY = pow(7, (R - 1) >> S, R)
DELTA = pow(7, 1 << S, R)
ZETA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23      # [RECALLED] halo2curves bn256 Fr::ZETA
ADVICE, FIXED, INSTANCE = 0, 1, 2
STREAM_ADVICE, STREAM_LOOKUP_PERMUTED, STREAM_PERM_Z, STREAM_LOOKUP_Z, STREAM_RANDOM_POLY = 0x11, 0x12, 0x13, 0x14, 0x15

_HERE = os.path.dirname(os.path.abspath(__file__))
_L = None


def lib():
    global _L
    if _L is None:
        _L = C.CDLL(os.path.join(_HERE, "libgl_oracle.so"))
    return _L


# ---- field / curve primitives ---------------------------------------------------------------------------------------------------
def inv(a):
    return pow(a, -1, R)


def limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def fft(a, inverse=False):
    """halo2 `best_fft` over the 2^k domain (EvaluationDomain::ifft includes the 1/n): natural order in and out"""
    n = len(a)
    if n == 1:
        return list(a)
    d = limbs(a)
    lib().orc_bn254_fr_ntt(d.ctypes.data_as(C.c_void_p), C.c_uint32(n.bit_length() - 1), C.c_int(1 if inverse else 0))
    return ints(d)


def pt_words(p):
    a = np.zeros(8, dtype=np.uint64)
    if p is not None:
        a[:4], a[4:] = limbs([p[0]])[0], limbs([p[1]])[0]
    return a


def pt_from(a):
    x, y = ints(a[:4])[0], ints(a[4:])[0]
    return None if x == 0 and y == 0 else (x, y)


def g1_mul(p, k):
    out = np.zeros(8, dtype=np.uint64)
    lib().orc_bn254_g1_mul(pt_words(p).ctypes.data_as(C.c_void_p), limbs([k % R]).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return pt_from(out)


def msm(points_words, scalars):
    """sum_i scalars[i] * points[i] (halo2 `best_multiexp`); points as an [n][8] uint64 array"""
    out = np.zeros(8, dtype=np.uint64)
    sc = limbs([s % R for s in scalars])
    lib().orc_bn254_g1_msm(points_words.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), C.c_size_t(len(scalars)), out.ctypes.data_as(C.c_void_p))
    return pt_from(out)


def eval_poly(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def kate_division(coeffs, z):
    """(p(X) - p(z)) / (X - z): halo2 `kate_division`, one coefficient shorter than p"""
    q = [0] * (len(coeffs) - 1)
    acc = 0
    for i in range(len(coeffs) - 1, 0, -1):
        acc = (coeffs[i] + acc * z) % R
        q[i - 1] = acc
    return q


# ---- Keccak-256 (the original Keccak padding 0x01, as Ethereum uses; not SHA3-256) -----------------------------------------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & _M64) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                r = _ROT[x][y]
                v = a[x][y]
                b[y][(2 * x + 3 * y) % 5] = ((v << r) | (v >> (64 - r))) & _M64 if r else v
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data):
    rate = 136
    p = bytearray(data)
    p.append(0x01)
    while len(p) % rate:
        p.append(0)
    p[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= struct.unpack_from("<Q", p, off + 8 * i)[0]
        a = _keccak_f(a)
    return b"".join(struct.pack("<Q", a[i % 5][i // 5]) for i in range(4))


class Keccak256Transcript:
    """[RECALLED] halo2_solidity_verifier::Keccak256Transcript (what test_utils.rs:73 constructs): points and scalars enter the running
    buffer as 32-byte big-endian words (x then y); a challenge = keccak256(buffer), as a big-endian integer mod r, and the digest becomes
    the new buffer (a 0x01 byte is appended when the buffer is exactly one previous digest, so two squeezes in a row differ)."""

    def __init__(self):
        self.buf, self.proof = bytearray(), bytearray()

    def common_scalar(self, s):
        self.buf += int(s).to_bytes(32, "big")

    def common_point(self, p):
        x, y = (0, 0) if p is None else p
        self.buf += x.to_bytes(32, "big") + y.to_bytes(32, "big")

    def write_scalar(self, s):
        self.common_scalar(s)
        self.proof += int(s).to_bytes(32, "big")

    def write_point(self, p):
        self.common_point(p)
        x, y = (0, 0) if p is None else p
        self.proof += x.to_bytes(32, "big") + y.to_bytes(32, "big")

    def squeeze_challenge(self):
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b"")
        h = keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R


# ---- ChaCha20 random scalars -------------------------------------------------------------------------------------------------------
def _chacha_block(key, counter, nonce):
    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & 0xFFFFFFFF
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key)) + [counter & 0xFFFFFFFF] + [v & 0xFFFFFFFF for v in nonce]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(x[i] + st[i]) & 0xFFFFFFFF for i in range(16)])


def random_fr(seed, stream, a, index):
    """scalar `index` of stream (stream, a) under the 32-byte seed: block counter = index, nonce = (stream, a, index >> 32)"""
    return int.from_bytes(_chacha_block(seed, index, (stream, a, index >> 32)), "little") % R


# ---- parameters and keys ---------------------------------------------------------------------------------------------------------
class Params:
    """ParamsKZG::<Bn256>::setup(k, rng) (verifier_api.rs:77) with the secret handed in: g[i] = [tau^i] G1, g_lagrange[i] = [L_i(tau)] G1"""

    def __init__(self, k, tau):
        self.k, self.n, self.tau = k, 1 << k, tau % R
        n = self.n
        self.omega = pow(ROOT_OF_UNITY, 1 << (S - k), R)
        g1 = (1, 2)
        self.g = np.stack([pt_words(g1_mul(g1, pow(self.tau, i, R))) for i in range(n)])
        tn = (pow(self.tau, n, R) - 1) * inv(n) % R
        self.g_lagrange = np.stack([pt_words(g1_mul(g1, tn * pow(self.omega, i, R) % R * inv((self.tau - pow(self.omega, i, R)) % R) % R)) for i in range(n)])

    def commit(self, coeffs):
        c = list(coeffs) + [0] * (self.n - len(coeffs))
        return msm(self.g, c)

    def commit_lagrange(self, values):
        return msm(self.g_lagrange, values)


class Domain:
    """poly::EvaluationDomain: the 2^k domain and the extended coset domain zeta * <omega_ext> of size 2^extended_k"""

    def __init__(self, k, degree):
        self.k, self.n = k, 1 << k
        self.quotient_poly_degree = degree - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k, self.ext_n = ek, 1 << ek
        self.omega = pow(ROOT_OF_UNITY, 1 << (S - k), R)
        self.omega_inv = inv(self.omega)
        self.ext_omega = pow(ROOT_OF_UNITY, 1 << (S - ek), R)

    def lagrange_to_coeff(self, values):
        return fft(values, inverse=True)

    def coeff_to_extended(self, coeffs):
        a = [c * pow(ZETA, i, R) % R for i, c in enumerate(coeffs)] + [0] * (self.ext_n - len(coeffs))
        return fft(a)

    def extended_to_coeff(self, values):
        a = fft(values, inverse=True)
        zi = inv(ZETA)
        return [c * pow(zi, i, R) % R for i, c in enumerate(a)]

    def rotate_omega(self, x, rot):
        return x * pow(self.omega if rot >= 0 else self.omega_inv, abs(rot), R) % R


class ProvingKey:
    pass


def keygen(params, cs, fixed_values, assembly):
    """keygen_vk + keygen_pk (verifier_api.rs:78-79) as far as the prover needs them: fixed polynomials and commitments, the permutation's
    sigma polynomials (sigma_j[i] = delta^(column) omega^(row) of the cell (j, i) maps to) and commitments, l_0 / l_last / l_active_row"""
    pk = ProvingKey()
    n = params.n
    dom = Domain(params.k, cs.degree())
    pk.domain, pk.cs = dom, cs
    bf = cs.blinding_factors()
    pk.usable = n - (bf + 1)
    assert cs.minimum_rows() <= n
    pk.fixed_values = [[v % R for v in col] for col in fixed_values]
    pk.fixed_polys = [dom.lagrange_to_coeff(col) for col in pk.fixed_values]
    pk.fixed_commitments = [params.commit_lagrange(col) for col in pk.fixed_values]
    m = len(cs.permutation)
    omega_pows = [pow(dom.omega, i, R) for i in range(n)]
    pk.sigma_values = []
    for j in range(m):
        col = []
        for i in range(n):
            cj, ci = (int(v) for v in assembly.mapping[j][i])
            col.append(pow(DELTA, cj, R) * omega_pows[ci] % R)
        pk.sigma_values.append(col)
    pk.sigma_polys = [dom.lagrange_to_coeff(c) for c in pk.sigma_values]
    pk.sigma_commitments = [params.commit_lagrange(c) for c in pk.sigma_values]
    l0 = [0] * n; l0[0] = 1
    ll = [0] * n; ll[pk.usable] = 1
    la = [1 if i < pk.usable else 0 for i in range(n)]          # 1 - (l_last + l_blind)
    pk.l0, pk.l_last, pk.l_active = (dom.lagrange_to_coeff(v) for v in (l0, ll, la))
    return pk


def compress(exprs, theta, query):
    acc = 0
    for e in exprs:
        acc = (acc * theta + e.evaluate(query)) % R
    return acc


# ---- create_proof ----------------------------------------------------------------------------------------------------------------
def create_proof(params, pk, advice, instances, seed, vk_digest, trace=None):
    """plonk::create_proof for one circuit instance, single phase.  advice: [num_advice][n] integers (rows >= usable are overwritten by
    blinding values), instances: per instance column the list of public values (zero-padded to n).  Returns the proof bytes.
    trace (a dict) receives the intermediate objects a parity test may want to compare."""
    cs, dom = pk.cs, pk.domain
    n, k, u = params.n, params.k, pk.usable
    bf = cs.blinding_factors()
    omega = dom.omega
    tr = Keccak256Transcript()
    T = trace if trace is not None else {}

    # -- vk, instances (KZG: instance columns are not committed, their values go into the transcript)
    tr.common_scalar(vk_digest)
    inst_values = []
    for col in instances:
        assert len(col) <= u
        for v in col:
            tr.common_scalar(v % R)
        inst_values.append([v % R for v in col] + [0] * (n - len(col)))
    inst_polys = [dom.lagrange_to_coeff(v) for v in inst_values]

    # -- advice: blind the unusable rows, commit in Lagrange form
    adv_values = []
    for c in range(cs.num_advice):
        col = [v % R for v in advice[c][:u]] + [random_fr(seed, STREAM_ADVICE, c, i) for i in range(u, n)]
        adv_values.append(col)
    adv_commitments = [params.commit_lagrange(col) for col in adv_values]
    for p in adv_commitments:
        tr.write_point(p)
    adv_polys = [dom.lagrange_to_coeff(v) for v in adv_values]
    T["advice_commitments"] = adv_commitments

    def row_query(i):
        def q(kind, qi):
            col, rot = cs.queries[kind][qi]
            src = adv_values if kind == ADVICE else (pk.fixed_values if kind == FIXED else inst_values)
            return src[col][(i + rot) % n]
        return q

    # -- lookups: compress with theta, permute (lookup::prover::Argument::commit_permuted)
    theta = tr.squeeze_challenge()
    lookups = []
    for li, (_, ins, tabs) in enumerate(cs.lookups):
        A = [compress(ins, theta, row_query(i)) for i in range(n)]
        Sv = [compress(tabs, theta, row_query(i)) for i in range(n)]
        Ap, Sp = permute_expression_pair(A, Sv, u)
        Ap += [random_fr(seed, STREAM_LOOKUP_PERMUTED, 2 * li, i) for i in range(u, n)]
        Sp += [random_fr(seed, STREAM_LOOKUP_PERMUTED, 2 * li + 1, i) for i in range(u, n)]
        ca, cs_ = params.commit_lagrange(Ap), params.commit_lagrange(Sp)
        tr.write_point(ca)
        tr.write_point(cs_)
        lookups.append(dict(A=A, S=Sv, Ap=Ap, Sp=Sp, Ap_poly=dom.lagrange_to_coeff(Ap), Sp_poly=dom.lagrange_to_coeff(Sp), ins=ins, tabs=tabs))
    T["theta"] = theta

    # -- permutation grand products (permutation::prover::Argument::commit)
    beta = tr.squeeze_challenge()
    gamma = tr.squeeze_challenge()
    T["beta"], T["gamma"] = beta, gamma
    cl = cs.chunk_len()

    def column_values(col):
        return adv_values[col.index] if col.kind == ADVICE else (pk.fixed_values[col.index] if col.kind == FIXED else inst_values[col.index])
    perm_sets = []
    last_z = 1
    omega_pows = [pow(omega, i, R) for i in range(n)]
    for s0 in range(0, len(cs.permutation), cl):
        cols = list(range(s0, min(s0 + cl, len(cs.permutation))))
        mod = [1] * n
        for j in cols:
            v, sg = column_values(cs.permutation[j]), pk.sigma_values[j]
            for i in range(n):
                mod[i] = mod[i] * ((beta * sg[i] + gamma + v[i]) % R) % R
        mod = [inv(x) for x in mod]
        for j in cols:
            v = column_values(cs.permutation[j])
            dj = pow(DELTA, j, R)
            for i in range(n):
                mod[i] = mod[i] * ((dj * omega_pows[i] % R * beta + gamma + v[i]) % R) % R
        z = [last_z]
        for i in range(n - 1):
            z.append(z[-1] * mod[i] % R)
        for i in range(u + 1, n):
            z[i] = random_fr(seed, STREAM_PERM_Z, len(perm_sets), i)
        last_z = z[u]
        c = params.commit_lagrange(z)
        tr.write_point(c)
        perm_sets.append(dict(cols=cols, z=z, poly=dom.lagrange_to_coeff(z)))
    T["perm_z"] = [s["z"] for s in perm_sets]

    # -- lookup grand products (lookup::prover::Permuted::commit_product)
    for li, lk in enumerate(lookups):
        den = [inv((lk["Ap"][i] + beta) * (lk["Sp"][i] + gamma) % R) for i in range(n)]
        prod = [den[i] * ((lk["A"][i] + beta) % R) % R * ((lk["S"][i] + gamma) % R) % R for i in range(n)]
        z = [1]
        for i in range(u):
            z.append(z[-1] * prod[i] % R)
        assert z[u] == 1, "lookup %d: the grand product does not close (input not in table?)" % li
        z += [random_fr(seed, STREAM_LOOKUP_Z, li, i) for i in range(u + 1, n)]
        tr.write_point(params.commit_lagrange(z))
        lk["z"], lk["z_poly"] = z, dom.lagrange_to_coeff(z)

    # -- vanishing argument, part 1: the random polynomial (vanishing::Argument::commit)
    random_poly = [random_fr(seed, STREAM_RANDOM_POLY, 0, i) for i in range(n)]
    tr.write_point(params.commit(random_poly))

    # -- the quotient h(X) on the extended domain (plonk/evaluation.rs evaluate_h), constraints folded with y in the verifier's order
    y = tr.squeeze_challenge()
    T["y"] = y
    N, step = dom.ext_n, dom.ext_n // n
    ext = dom.coeff_to_extended
    adv_e, fix_e, inst_e = [ext(p) for p in adv_polys], [ext(p) for p in pk.fixed_polys], [ext(p) for p in inst_polys]
    sig_e = [ext(p) for p in pk.sigma_polys]
    l0_e, ll_e, la_e = ext(pk.l0), ext(pk.l_last), ext(pk.l_active)
    for s in perm_sets:
        s["e"] = ext(s["poly"])
    for lk in lookups:
        lk["Ap_e"], lk["Sp_e"], lk["z_e"] = ext(lk["Ap_poly"]), ext(lk["Sp_poly"]), ext(lk["z_poly"])
    last_rot = -(bf + 1)
    xs, x = [], ZETA
    for _ in range(N):
        xs.append(x)
        x = x * dom.ext_omega % R
    h = [0] * N
    gate_polys = cs.all_gate_polys()

    def col_e(col):
        return adv_e[col.index] if col.kind == ADVICE else (fix_e[col.index] if col.kind == FIXED else inst_e[col.index])
    for i in range(N):
        def q(kind, qi, i=i):
            col, rot = cs.queries[kind][qi]
            src = adv_e if kind == ADVICE else (fix_e if kind == FIXED else inst_e)
            return src[col][(i + rot * step) % N]
        acc = 0
        for p in gate_polys:
            acc = (acc * y + p.evaluate(q)) % R
        if perm_sets:
            nxt, lst = (i + step) % N, (i + last_rot * step) % N
            acc = (acc * y + l0_e[i] * (1 - perm_sets[0]["e"][i])) % R
            zl = perm_sets[-1]["e"][i]
            acc = (acc * y + ll_e[i] * (zl * zl - zl)) % R
            for s in range(1, len(perm_sets)):
                acc = (acc * y + l0_e[i] * (perm_sets[s]["e"][i] - perm_sets[s - 1]["e"][lst])) % R
            for s in perm_sets:
                left, right = s["e"][nxt], s["e"][i]
                cur = pow(DELTA, s["cols"][0], R) * beta % R * xs[i] % R
                for j in s["cols"]:
                    v = col_e(cs.permutation[j])[i]
                    left = left * ((v + beta * sig_e[j][i] + gamma) % R) % R
                    right = right * ((v + cur + gamma) % R) % R
                    cur = cur * DELTA % R
                acc = (acc * y + (left - right) * la_e[i]) % R
        for lk in lookups:
            nxt, prv = (i + step) % N, (i - step) % N
            a_in, s_in = compress(lk["ins"], theta, q), compress(lk["tabs"], theta, q)
            z_i, ap, sp = lk["z_e"][i], lk["Ap_e"][i], lk["Sp_e"][i]
            acc = (acc * y + l0_e[i] * (1 - z_i)) % R
            acc = (acc * y + ll_e[i] * (z_i * z_i - z_i)) % R
            acc = (acc * y + (lk["z_e"][nxt] * (ap + beta) % R * (sp + gamma) - z_i * (a_in + beta) % R * (s_in + gamma)) % R * la_e[i]) % R
            acc = (acc * y + l0_e[i] * (ap - sp)) % R
            acc = (acc * y + (ap - sp) * (ap - lk["Ap_e"][prv]) % R * la_e[i]) % R
        h[i] = acc

    # -- vanishing argument, part 2: divide by X^n - 1, back to coefficients, split into pieces of n, commit (vanishing::Committed::construct)
    t_inv = [inv((pow(xs[j], n, R) - 1) % R) for j in range(step)]           # (zeta omega_ext^j)^n has period ext_n / n in j
    h = [h[i] * t_inv[i % step] % R for i in range(N)]
    h_coeffs = dom.extended_to_coeff(h)
    n_pieces = dom.quotient_poly_degree
    assert all(c == 0 for c in h_coeffs[n * n_pieces:]), "constraints not satisfied: the quotient has a remainder"
    pieces = [h_coeffs[i * n:(i + 1) * n] for i in range(n_pieces)]
    for p in pieces:
        tr.write_point(params.commit(p))
    T["h_pieces"] = pieces

    # -- evaluations at x
    x = tr.squeeze_challenge()
    T["x"] = x
    xn = pow(x, n, R)
    for col, rot in cs.queries[ADVICE]:
        tr.write_scalar(eval_poly(adv_polys[col], dom.rotate_omega(x, rot)))
    for col, rot in cs.queries[FIXED]:
        tr.write_scalar(eval_poly(pk.fixed_polys[col], dom.rotate_omega(x, rot)))
    h_poly = [0] * n                                                            # sum_i x^(n i) h_i(X)
    for p in reversed(pieces):
        h_poly = [(a * xn + b) % R for a, b in zip(h_poly, p)]
    tr.write_scalar(eval_poly(random_poly, x))
    for p in pk.sigma_polys:
        tr.write_scalar(eval_poly(p, x))
    x_next, x_last, x_inv = dom.rotate_omega(x, 1), dom.rotate_omega(x, last_rot), dom.rotate_omega(x, -1)
    for si, s in enumerate(perm_sets):
        tr.write_scalar(eval_poly(s["poly"], x))
        tr.write_scalar(eval_poly(s["poly"], x_next))
        if si + 1 < len(perm_sets):
            tr.write_scalar(eval_poly(s["poly"], x_last))
    for lk in lookups:
        tr.write_scalar(eval_poly(lk["z_poly"], x))
        tr.write_scalar(eval_poly(lk["z_poly"], x_next))
        tr.write_scalar(eval_poly(lk["Ap_poly"], x))
        tr.write_scalar(eval_poly(lk["Ap_poly"], x_inv))
        tr.write_scalar(eval_poly(lk["Sp_poly"], x))

    # -- the opening queries, in create_proof's order: (polynomial, point)
    queries = []
    for col, rot in cs.queries[ADVICE]:
        queries.append((("advice", col), adv_polys[col], dom.rotate_omega(x, rot)))
    for si, s in enumerate(perm_sets):
        queries.append((("perm_z", si), s["poly"], x))
        queries.append((("perm_z", si), s["poly"], x_next))
    for si in range(len(perm_sets) - 2, -1, -1):
        queries.append((("perm_z", si), perm_sets[si]["poly"], x_last))
    for li, lk in enumerate(lookups):
        queries.append((("lk_z", li), lk["z_poly"], x))
        queries.append((("lk_a", li), lk["Ap_poly"], x))
        queries.append((("lk_s", li), lk["Sp_poly"], x))
        queries.append((("lk_a", li), lk["Ap_poly"], x_inv))
        queries.append((("lk_z", li), lk["z_poly"], x_next))
    for col, rot in cs.queries[FIXED]:
        queries.append((("fixed", col), pk.fixed_polys[col], dom.rotate_omega(x, rot)))
    for j, p in enumerate(pk.sigma_polys):
        queries.append((("sigma", j), p, x))
    queries.append((("h",), h_poly, x))
    queries.append((("random",), random_poly, x))
    shplonk_open(params, tr, queries, T)
    return bytes(tr.proof)


def permute_expression_pair(A, Sv, usable):
    """lookup::prover::permute_expression_pair: the input sorted; the table rearranged so that every first occurrence of an input value
    faces itself and the remaining rows take the unused table values (ascending) from the LAST repeated row backwards"""
    a_sorted = sorted(A[:usable])
    leftover = {}
    for v in Sv[:usable]:
        leftover[v] = leftover.get(v, 0) + 1
    table = [0] * usable
    repeated = []
    for row, v in enumerate(a_sorted):
        if row == 0 or v != a_sorted[row - 1]:
            table[row] = v
            assert leftover.get(v, 0) > 0, "lookup input %d is not in the table" % v
            leftover[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(leftover):
        for _ in range(leftover[v]):
            table[repeated.pop()] = v
    assert not repeated
    return a_sorted, table


def lagrange_interpolate(points, evals):
    """coefficients of the polynomial of degree < len(points) through (points[i], evals[i])"""
    m = len(points)
    out = [0] * m
    for i in range(m):
        num, den = [1], 1
        for j in range(m):
            if j != i:
                num = [(a - points[j] * b) % R for a, b in zip([0] + num, num + [0])]
                den = den * (points[i] - points[j]) % R
        c = evals[i] * inv(den) % R
        for t in range(m):
            out[t] = (out[t] + c * num[t]) % R
    return out


def intermediate_sets(queries):
    """[RECALLED] poly/kzg/multiopen/shplonk.rs construct_intermediate_sets: polynomials in first-appearance order, each with the sorted set
    of its points; rotation sets = the distinct point sets in first-appearance order, each with its polynomials (in order) and their
    evaluations listed in the set's (sorted) point order; the super point set = all points, sorted."""
    order, by_key = [], {}
    for key, poly, pt in queries:
        if key not in by_key:
            by_key[key] = dict(poly=poly, points=set())
            order.append(key)
        by_key[key]["points"].add(pt)
    sets, index = [], {}
    for key in order:
        pts = tuple(sorted(by_key[key]["points"]))
        if pts not in index:
            index[pts] = len(sets)
            sets.append(dict(points=list(pts), keys=[], polys=[]))
        s = sets[index[pts]]
        s["keys"].append(key)
        s["polys"].append(by_key[key]["poly"])
    super_points = sorted({pt for _, _, pt in queries})
    return sets, super_points


def shplonk_open(params, tr, queries, T):
    """ProverSHPLONK::create_proof (BDFG21, section 4): y, v; h = sum_i v^i (sum_j y^j (p_ij - r_ij)) / Z_i, committed; u; the linearisation
    L(X) = sum_i v^i Z_{T \\ S_i}(u) sum_j y^j (p_ij(X) - r_ij(u)) - Z_T(u) h(X), normalised by Z_{T \\ S_0}(u); its quotient by (X - u), committed"""
    n = params.n
    y = tr.squeeze_challenge()
    v = tr.squeeze_challenge()
    sets, super_points = intermediate_sets(queries)
    h = [0] * n
    vi = 1
    for s in sets:
        s["evals"] = [[eval_poly(p, pt) for pt in s["points"]] for p in s["polys"]]
        s["r"] = [lagrange_interpolate(s["points"], ev) for ev in s["evals"]]
        nx, yj = [0] * n, 1
        for p, r in zip(s["polys"], s["r"]):
            pr = list(p) + [0] * (n - len(p))
            for t, c in enumerate(r):
                pr[t] = (pr[t] - c) % R
            nx = [(a + yj * b) % R for a, b in zip(nx, pr)]
            yj = yj * y % R
        for pt in s["points"]:
            nx = kate_division(nx, pt) + [0]
        h = [(a + vi * b) % R for a, b in zip(h, nx)]
        vi = vi * v % R
    tr.write_point(params.commit(h))
    u = tr.squeeze_challenge()
    zt = 1
    for pt in super_points:
        zt = zt * (u - pt) % R
    lx, vi, z0 = [0] * n, 1, None
    for s in sets:
        zi = 1
        for pt in super_points:
            if pt not in s["points"]:
                zi = zi * (u - pt) % R
        if z0 is None:
            z0 = zi
        inner, yj = [0] * n, 1
        for p, r in zip(s["polys"], s["r"]):
            pr = list(p) + [0] * (n - len(p))
            pr[0] = (pr[0] - eval_poly(r, u)) % R
            inner = [(a + yj * b) % R for a, b in zip(inner, pr)]
            yj = yj * y % R
        lx = [(a + vi * zi % R * b) % R for a, b in zip(lx, inner)]
        vi = vi * v % R
    lx = [(a - zt * b) % R for a, b in zip(lx, h)]
    assert eval_poly(lx, u) == 0
    z0i = inv(z0)
    lx = [a * z0i % R for a in lx]
    tr.write_point(params.commit(kate_division(lx, u)))
    T["shplonk"] = dict(y=y, v=v, u=u, n_sets=len(sets), n_points=len(super_points))
