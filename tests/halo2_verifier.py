"""TEST INFRASTRUCTURE: a restatement of halo2_proofs' `verify_proof::<_, VerifierSHPLONK<_>, _, Keccak256Transcript, SingleStrategy>` -- the check
the reference runs on every finalisation proof (/root/reference/src/plonky2_verifier/chip/native_chip/test_utils.rs:82-93) -- written
independently of the prover restatement (oracle/halo2_model.py) and of the product: it shares with them only the circuit description
(ConstraintSystem / Expression trees of stark-verifier_amd/halo2.py), never prover code.  Group arithmetic: tests/pymodel_bn254_curve.py (affine,
big integers).  The final pairing check e(L + u W', [1]_2) = e(W', [tau]_2) is evaluated in the exponent: the SRS secret tau is known to
the tests, so it reads L + u W' = tau W' in G1.

Follows the published verifier (plonk/verifier.rs, plonk/{permutation,lookup,vanishing}/verifier.rs, poly/kzg/multiopen/shplonk/verifier.rs):
read commitments and challenges in the prover's order, read the evaluations, recompute the quotient's evaluation from the gate / permutation /
lookup expressions at x, and verify one SHPLONK opening of every (commitment, point, evaluation) triple."""
import hashlib  # noqa: F401  (kept out of the transcript: SHA3-256 is not Keccak-256)

import pymodel_bn254_curve as pm

R = pm.R
ADVICE, FIXED, INSTANCE = 0, 1, 2


class VerifyError(Exception):
    pass


def keccak256(data):
    """Keccak-256 (pad 0x01): bit-sliced lanes as plain integers, separate from the prover restatement's implementation"""
    rc = []
    lfsr = 1
    for _ in range(24):
        c = 0
        for j in range(7):
            if lfsr & 1:
                c ^= 1 << ((1 << j) - 1)
            lfsr = ((lfsr << 1) ^ (0x71 if lfsr & 0x80 else 0)) & 0xFF
        rc.append(c)
    rot = {}
    x, y = 1, 0
    for t in range(24):
        rot[(x, y)] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    rot[(0, 0)] = 0
    m = (1 << 64) - 1
    rate = 136
    pad = rate - len(data) % rate
    msg = bytes(data) + (b"\x81" if pad == 1 else b"\x01" + b"\x00" * (pad - 2) + b"\x80")
    st = [0] * 25                                     # st[x + 5 y]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        for r in range(24):
            c = [st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20] for x in range(5)]
            d = [c[(x + 4) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & m) for x in range(5)]
            st = [st[i] ^ d[i % 5] for i in range(25)]
            b = [0] * 25
            for x in range(5):
                for y in range(5):
                    v, k = st[x + 5 * y], rot[(x, y)]
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = ((v << k) | (v >> (64 - k))) & m if k else v
            st = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & m & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]
            st[0] ^= rc[r]
    return b"".join(st[i].to_bytes(8, "little") for i in range(4))


class TranscriptReader:
    def __init__(self, proof):
        self.proof, self.pos, self.buf = bytes(proof), 0, bytearray()

    def common_scalar(self, s):
        self.buf += int(s).to_bytes(32, "big")

    def read_point(self):
        raw = self.proof[self.pos:self.pos + 64]
        if len(raw) != 64:
            raise VerifyError("proof too short (point)")
        self.pos += 64
        x, y = int.from_bytes(raw[:32], "big"), int.from_bytes(raw[32:], "big")
        if x >= pm.Q or y >= pm.Q:
            raise VerifyError("point coordinate not canonical")
        if (x, y) != (0, 0) and (y * y - x * x * x - 3) % pm.Q:
            raise VerifyError("point not on the curve")
        self.buf += raw
        return None if (x, y) == (0, 0) else (x, y)

    def read_scalar(self):
        raw = self.proof[self.pos:self.pos + 32]
        if len(raw) != 32:
            raise VerifyError("proof too short (scalar)")
        self.pos += 32
        v = int.from_bytes(raw, "big")
        if v >= R:
            raise VerifyError("scalar not canonical")
        self.buf += raw
        return v

    def squeeze_challenge(self):
        h = keccak256(bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b""))
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R


def lagrange_evals(k, x, rows):
    """{i: l_i(x)} for the requested rows of the 2^k domain: l_i(x) = (x^n - 1) / n * omega^i / (x - omega^i)"""
    n = 1 << k
    w = pm.omega(k)
    base = (pow(x, n, R) - 1) * pow(n, -1, R) % R
    out = {}
    for i in rows:
        wi = pow(w, i % n, R)
        out[i] = base * wi % R * pow((x - wi) % R, -1, R) % R
    return out


def verify(k, cs, vk, instances, proof, tau):
    """vk: dict(digest, fixed_commitments, sigma_commitments) with points as (x, y) integer pairs / None.  Raises VerifyError."""
    n = 1 << k
    bf = cs.blinding_factors()
    u = n - (bf + 1)
    omega = pm.omega(k)
    omega_inv = pow(omega, -1, R)
    tr = TranscriptReader(proof)
    tr.common_scalar(vk["digest"])
    for col in instances:
        if len(col) > u:
            raise VerifyError("too many instance values")
        for v in col:
            tr.common_scalar(v % R)
    advice_c = [tr.read_point() for _ in range(cs.num_advice)]
    theta = tr.squeeze_challenge()
    lookups_c = [[tr.read_point(), tr.read_point()] for _ in cs.lookups]           # permuted input, permuted table
    beta = tr.squeeze_challenge()
    gamma = tr.squeeze_challenge()
    cl = cs.chunk_len()
    n_sets = (len(cs.permutation) + cl - 1) // cl if cs.permutation else 0
    perm_c = [tr.read_point() for _ in range(n_sets)]
    for lc in lookups_c:
        lc.append(tr.read_point())                                                 # product
    random_c = tr.read_point()
    y = tr.squeeze_challenge()
    n_pieces = cs.degree() - 1
    h_c = [tr.read_point() for _ in range(n_pieces)]
    x = tr.squeeze_challenge()
    adv_ev = [tr.read_scalar() for _ in cs.queries[ADVICE]]
    fix_ev = [tr.read_scalar() for _ in cs.queries[FIXED]]
    random_ev = tr.read_scalar()
    sigma_ev = [tr.read_scalar() for _ in cs.permutation]
    perm_ev = []
    for s in range(n_sets):
        e = dict(z=tr.read_scalar(), z_next=tr.read_scalar())
        if s + 1 < n_sets:
            e["z_last"] = tr.read_scalar()
        perm_ev.append(e)
    lk_ev = [dict(z=tr.read_scalar(), z_next=tr.read_scalar(), a=tr.read_scalar(), a_inv=tr.read_scalar(), s=tr.read_scalar()) for _ in cs.lookups]

    def rot(point, r):
        return point * pow(omega if r >= 0 else omega_inv, abs(r), R) % R
    # instance evaluations from the public values (KZG: the verifier evaluates the Lagrange form itself)
    inst_ev = []
    for col, r_ in cs.queries[INSTANCE]:
        vals = instances[col]
        le = lagrange_evals(k, rot(x, r_), range(len(vals)))
        inst_ev.append(sum(v % R * le[i] for i, v in enumerate(vals)) % R)
    xn = pow(x, n, R)
    le = lagrange_evals(k, x, [0] + list(range(u, n)))
    l_0, l_last = le[0], le[u]
    l_blind = sum(le[i] for i in range(u + 1, n)) % R
    l_active = (1 - l_last - l_blind) % R

    def q(kind, qi):
        return (adv_ev if kind == ADVICE else (fix_ev if kind == FIXED else inst_ev))[qi]

    def column_eval(col):                      # the permutation argument reads every equality column at rotation 0
        qs = cs.queries[col.kind]
        if (col.index, 0) not in qs:
            raise VerifyError("equality column %r is not queried at the current rotation" % (col,))
        return q(col.kind, qs.index((col.index, 0)))
    # the quotient's evaluation, folded with y in the same order as the prover's h(X)
    acc = 0
    for p in cs.all_gate_polys():
        acc = (acc * y + p.evaluate(q)) % R
    if n_sets:
        acc = (acc * y + l_0 * (1 - perm_ev[0]["z"])) % R
        zl = perm_ev[-1]["z"]
        acc = (acc * y + l_last * (zl * zl - zl)) % R
        for s in range(1, n_sets):
            acc = (acc * y + l_0 * (perm_ev[s]["z"] - perm_ev[s - 1]["z_last"])) % R
        for s in range(n_sets):
            left, right = perm_ev[s]["z_next"], perm_ev[s]["z"]
            delta = pow(7, 1 << 28, R)                          # Fr::DELTA
            cur = pow(delta, s * cl, R) * beta % R * x % R
            for j in range(s * cl, min((s + 1) * cl, len(cs.permutation))):
                v = column_eval(cs.permutation[j])
                left = left * ((v + beta * sigma_ev[j] + gamma) % R) % R
                right = right * ((v + cur + gamma) % R) % R
                cur = cur * delta % R
            acc = (acc * y + (left - right) * l_active) % R
    for (name, ins, tabs), ev in zip(cs.lookups, lk_ev):
        a_in = 0
        for e in ins:
            a_in = (a_in * theta + e.evaluate(q)) % R
        s_in = 0
        for e in tabs:
            s_in = (s_in * theta + e.evaluate(q)) % R
        acc = (acc * y + l_0 * (1 - ev["z"])) % R
        acc = (acc * y + l_last * (ev["z"] * ev["z"] - ev["z"])) % R
        acc = (acc * y + (ev["z_next"] * (ev["a"] + beta) % R * (ev["s"] + gamma) - ev["z"] * (a_in + beta) % R * (s_in + gamma)) % R * l_active) % R
        acc = (acc * y + l_0 * (ev["a"] - ev["s"])) % R
        acc = (acc * y + (ev["a"] - ev["s"]) * (ev["a"] - ev["a_inv"]) % R * l_active) % R
    if xn == 1:
        raise VerifyError("x in the domain")
    h_eval = acc * pow((xn - 1) % R, -1, R) % R
    h_commit = None
    for c in reversed(h_c):
        h_commit = pm.add(pm.mul(h_commit, xn), c)
    # (commitment, point, evaluation) in the prover's query order
    x_next, x_last, x_inv = rot(x, 1), rot(x, -(bf + 1)), rot(x, -1)
    queries = []
    for qi, (col, r_) in enumerate(cs.queries[ADVICE]):
        queries.append((("advice", col), advice_c[col], rot(x, r_), adv_ev[qi]))
    for s in range(n_sets):
        queries.append((("perm_z", s), perm_c[s], x, perm_ev[s]["z"]))
        queries.append((("perm_z", s), perm_c[s], x_next, perm_ev[s]["z_next"]))
    for s in range(n_sets - 2, -1, -1):
        queries.append((("perm_z", s), perm_c[s], x_last, perm_ev[s]["z_last"]))
    for li, (lc, ev) in enumerate(zip(lookups_c, lk_ev)):
        queries.append((("lk_z", li), lc[2], x, ev["z"]))
        queries.append((("lk_a", li), lc[0], x, ev["a"]))
        queries.append((("lk_s", li), lc[1], x, ev["s"]))
        queries.append((("lk_a", li), lc[0], x_inv, ev["a_inv"]))
        queries.append((("lk_z", li), lc[2], x_next, ev["z_next"]))
    for qi, (col, r_) in enumerate(cs.queries[FIXED]):
        queries.append((("fixed", col), vk["fixed_commitments"][col], rot(x, r_), fix_ev[qi]))
    for j in range(len(cs.permutation)):
        queries.append((("sigma", j), vk["sigma_commitments"][j], x, sigma_ev[j]))
    queries.append((("h",), h_commit, x, h_eval))
    queries.append((("random",), random_c, x, random_ev))
    shplonk_verify(tr, queries, tau)
    if tr.pos != len(tr.proof):
        raise VerifyError("trailing bytes in the proof")
    return True


def interpolate_eval(points, evals, at):
    """value at `at` of the polynomial of degree < len(points) through (points[i], evals[i])"""
    total = 0
    for i, (pi, ei) in enumerate(zip(points, evals)):
        num, den = 1, 1
        for j, pj in enumerate(points):
            if j != i:
                num = num * (at - pj) % R
                den = den * (pi - pj) % R
        total = (total + ei * num % R * pow(den, -1, R)) % R
    return total


def shplonk_verify(tr, queries, tau):
    y = tr.squeeze_challenge()
    v = tr.squeeze_challenge()
    h1 = tr.read_point()
    u = tr.squeeze_challenge()
    h2 = tr.read_point()
    # group the commitments by their point sets, as the prover does (first-appearance order; points sorted)
    order, info = [], {}
    for key, c, pt, ev in queries:
        if key not in info:
            info[key] = dict(c=c, evals={})
            order.append(key)
        if pt in info[key]["evals"] and info[key]["evals"][pt] != ev:
            raise VerifyError("two different evaluations claimed for one (commitment, point)")
        info[key]["evals"][pt] = ev
    sets, index = [], {}
    for key in order:
        pts = tuple(sorted(info[key]["evals"]))
        if pts not in index:
            index[pts] = len(sets)
            sets.append((pts, []))
        sets[index[pts]][1].append(key)
    all_points = sorted({pt for _, _, pt, _ in queries})
    zt = 1
    for pt in all_points:
        zt = zt * (u - pt) % R
    outer_c, outer_r, vi, z0 = None, 0, 1, None
    for pts, keys in sets:
        zi = 1
        for pt in all_points:
            if pt not in pts:
                zi = zi * (u - pt) % R
        if z0 is None:
            z0 = zi
        inner_c, inner_r, yj = None, 0, 1
        for key in keys:
            r_u = interpolate_eval(list(pts), [info[key]["evals"][pt] for pt in pts], u)
            inner_c = pm.add(inner_c, pm.mul(info[key]["c"], yj))
            inner_r = (inner_r + yj * r_u) % R
            yj = yj * y % R
        scale = vi * zi % R
        outer_c = pm.add(outer_c, pm.mul(inner_c, scale))
        outer_r = (outer_r + scale * inner_r) % R
        vi = vi * v % R
    z0i = pow(z0, -1, R)
    # L = (sum_i v^i z_i (C_i - [r_i] G) - [Z_T(u)] h1) / z_0 ;   e(L + u h2, G2) = e(h2, tau G2)  <=>  L = (tau - u) h2
    left = pm.add(outer_c, pm.mul(pm.G, (-outer_r) % R))
    left = pm.add(left, pm.mul(h1, (-zt) % R))
    left = pm.mul(left, z0i)
    right = pm.mul(h2, (tau - u) % R)
    if left != right:
        raise VerifyError("SHPLONK opening check failed")
