"""Restatement of the reference's plonky2 verifier (TEST INFRASTRUCTURE): the checks the Halo2 circuit
in src/plonky2_verifier/chip performs on a proof, in plain Python over big integers.

  get_challenges                 chip/plonk/plonk_verifier_chip.rs:55-154
  verify_proof_with_challenges   chip/plonk/plonk_verifier_chip.rs:156-242
  eval_vanishing_poly            chip/plonk/vanishing_poly.rs:18-218
  gate filter / evaluators       chip/plonk/gates/mod.rs:87-132, gates/{noop,constant,public_input,base_sum,
                                 arithmetic,poseidon}.rs
  FRI                            chip/fri_chip.rs:58-376, Merkle paths chip/merkle_proof_chip.rs:39-87
  sponge / transcript            chip/hasher_chip.rs:48-148
A proof produced by the GPU prover must pass every one of these equations.  Hashing goes through the
CPU oracle (also test infrastructure)."""
import os
import sys

import numpy as np

import pymodel as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon_tables as gpt  # noqa: E402

P = pm.P
(NOOP, CONSTANT, PUBLIC_INPUT, BASE_SUM, POSEIDON, ARITHMETIC, ARITHMETIC_EXT, MUL_EXT, POSEIDON_MDS, RANDOM_ACCESS,
 REDUCING, REDUCING_EXT) = range(12)
UNUSED_SELECTOR = 0xFFFFFFFF
_RC = gpt.load_rc()
_TB = gpt.derive(_RC)


class VerifyError(AssertionError):
    pass


def ext(x):
    return (int(x[0]) % P, int(x[1]) % P)


def base(x):
    return (int(x) % P, 0)


E0, E1 = (0, 0), (1, 0)
add, sub, mul = pm.ext_add, pm.ext_sub, pm.ext_mul


def reduce_with_powers(terms, a):
    """sum_i a^i t_i (goldilocks_extension_chip.rs:331-342)"""
    acc = E0
    for t in reversed(terms):
        acc = add(mul(acc, a), t)
    return acc


def ext_pow(a, e):
    r = E1
    while e:
        if e & 1:
            r = mul(r, a)
        a = mul(a, a)
        e >>= 1
    return r


class Transcript:
    def __init__(self, orc, hasher=0):
        self.orc, self.ch = orc, orc.challenger(hasher)

    def observe(self, elems):
        self.orc.observe(self.ch, np.asarray(elems, dtype=np.uint64).reshape(-1))

    def squeeze(self, n):
        return [self.orc.squeeze(self.ch) for _ in range(n)]


# ---- gate evaluators in the extension field ------------------------------------------------------------
def _sbox(x):
    x2 = mul(x, x)
    x4 = mul(x2, x2)
    return mul(mul(x, x2), x4)


def _mds(s):
    m = gpt.mds_matrix()
    return [reduce_sum([mul(base(m[r][c]), s[c]) for c in range(12)]) for r in range(12)]


def reduce_sum(xs):
    acc = E0
    for x in xs:
        acc = add(acc, x)
    return acc


def eval_poseidon(consts, w, pi_hash):
    c = []
    swap = w[24]
    c.append(sub(mul(swap, swap), swap))
    st = [None] * 12
    for i in range(4):
        lhs, rhs, delta = w[i], w[i + 4], w[25 + i]
        c.append(sub(mul(swap, sub(rhs, lhs)), delta))
        st[i] = add(lhs, delta)
        st[i + 4] = sub(rhs, delta)
    for i in range(8, 12):
        st[i] = w[i]
    rnd = 0
    for r in range(4):
        st = [add(x, base(_RC[rnd][i])) for i, x in enumerate(st)]
        if r != 0:
            for i in range(12):
                sin = w[29 + 12 * (r - 1) + i]
                c.append(sub(st[i], sin))
                st[i] = sin
        st = _mds([_sbox(x) for x in st])
        rnd += 1
    st = [add(x, base(_TB["first"][i])) for i, x in enumerate(st)]
    st = [st[0]] + [reduce_sum([mul(base(_TB["init"][r - 1][cc - 1]), st[r]) for r in range(1, 12)]) for cc in range(1, 12)]
    for r in range(22):
        sin = w[65 + r]
        c.append(sub(st[0], sin))
        s0 = _sbox(sin)
        if r < 21:
            s0 = add(s0, base(_TB["post"][r]))
        d = add(mul(base(_TB["m00"]), s0), reduce_sum([mul(base(_TB["w_hats"][r][i - 1]), st[i]) for i in range(1, 12)]))
        st = [d] + [add(st[i], mul(base(_TB["vs"][r][i - 1]), s0)) for i in range(1, 12)]
    rnd += 22
    for r in range(4):
        st = [add(x, base(_RC[rnd][i])) for i, x in enumerate(st)]
        for i in range(12):
            sin = w[87 + 12 * r + i]
            c.append(sub(st[i], sin))
            st[i] = sin
        st = _mds([_sbox(x) for x in st])
        rnd += 1
    for i in range(12):
        c.append(sub(st[i], w[12 + i]))
    assert len(c) == 123
    return c


def eval_gate(gate, consts, w, pi_hash):
    t, p = gate
    if t == NOOP:
        return []
    if t == CONSTANT:
        return [sub(consts[i], w[i]) for i in range(p)]
    if t == PUBLIC_INPUT:
        return [sub(w[i], base(pi_hash[i])) for i in range(4)]
    if t == BASE_SUM:
        limbs = w[1:1 + p]
        out = [sub(reduce_with_powers(limbs, base(2)), w[0])]
        return out + [sub(mul(l, l), l) for l in limbs]
    if t == ARITHMETIC:
        return [sub(w[4 * i + 3], add(mul(mul(w[4 * i], w[4 * i + 1]), consts[0]), mul(w[4 * i + 2], consts[1]))) for i in range(p)]
    if t == POSEIDON:
        return eval_poseidon(consts, w, pi_hash)
    # ---- gates over the extension algebra (goldilocks_extension_algebra_chip.rs:112-146): an element is a
    # pair of wires (A0, A1), product (A0 B0 + 7 A1 B1, A0 B1 + A1 B0) with component arithmetic in K
    def alg(j):
        return (w[j], w[j + 1])

    def amul(a, b):
        return (add(mul(a[0], b[0]), mul(base(7), mul(a[1], b[1]))), add(mul(a[0], b[1]), mul(a[1], b[0])))

    def aadd(a, b):
        return (add(a[0], b[0]), add(a[1], b[1]))

    def asub(a, b):
        return (sub(a[0], b[0]), sub(a[1], b[1]))

    def ascal(c, a):
        return (mul(c, a[0]), mul(c, a[1]))
    if t == ARITHMETIC_EXT:
        out = []
        for i in range(p):
            comp = aadd(ascal(consts[0], amul(alg(8 * i), alg(8 * i + 2))), ascal(consts[1], alg(8 * i + 4)))
            out += list(asub(alg(8 * i + 6), comp))
        return out
    if t == MUL_EXT:
        out = []
        for i in range(p):
            out += list(asub(alg(6 * i + 4), ascal(consts[0], amul(alg(6 * i), alg(6 * i + 2)))))
        return out
    if t == POSEIDON_MDS:
        out = []
        for r in range(12):
            acc = (E0, E0)
            for i in range(12):
                acc = aadd(acc, ascal(base(gpt.CIRC[i]), alg(2 * ((i + r) % 12))))
            acc = aadd(acc, ascal(base(gpt.DIAG[r]), alg(2 * r)))
            out += list(asub(alg(2 * (12 + r)), acc))
        return out
    if t == RANDOM_ACCESS:
        bits, copies, extra = p & 0xFF, (p >> 8) & 0xFF, (p >> 16) & 0xFF
        vec = 1 << bits
        routed = (2 + vec) * copies + extra
        out = []
        for c in range(copies):
            b0 = (2 + vec) * c
            bl = [w[routed + c * bits + i] for i in range(bits)]
            out += [sub(mul(b, b), b) for b in bl]
            out.append(sub(reduce_with_powers(bl, base(2)), w[b0]))
            items = [w[b0 + 2 + i] for i in range(vec)]
            for b in bl:
                items = [add(mul(b, sub(items[2 * k + 1], items[2 * k])), items[2 * k]) for k in range(len(items) // 2)]
            out.append(sub(items[0], w[b0 + 1]))
        out += [sub(consts[i], w[(2 + vec) * copies + i]) for i in range(extra)]
        return out
    if t in (REDUCING, REDUCING_EXT):
        isext = t == REDUCING_EXT
        alpha, acc = alg(2), alg(4)
        start_accs = 6 + (2 * p if isext else p)
        out = []
        for i in range(p):
            coeff = alg(6 + 2 * i) if isext else (w[6 + i], E0)
            acc_i = alg(0) if i == p - 1 else alg(start_accs + 2 * i)
            out += list(asub(aadd(amul(acc, alpha), coeff), acc_i))
            acc = acc_i
        return out
    raise VerifyError("unknown gate")


def eval_vanishing_poly(cd, x, x_pow_n, op, pi_hash, betas, gammas, alphas):
    n = 1 << cd["degree_bits"]
    consts, wires = op["constants"], op["wires"]
    # gate constraints with filters (gates/mod.rs:87-132)
    allc = [E0] * cd["num_gate_constraints"]
    for gi, gate in enumerate(cd["gates"]):
        sel = cd["selector_indices"][gi]
        lo, hi = cd["groups"][sel]
        f = consts[sel]
        filt = E1
        for k in [k for k in range(lo, hi) if k != gi] + ([UNUSED_SELECTOR] if cd["num_selectors"] > 1 else []):
            filt = mul(filt, sub(base(k), f))
        for k, c in enumerate(eval_gate(gate, consts[cd["num_selectors"]:], wires, pi_hash)):
            allc[k] = add(allc[k], mul(filt, c))
    # L0(x) = (x^n - 1) / (n (x - 1))
    l0 = mul(sub(x_pow_n, E1), pm.ext_inv(sub(mul(base(n), x), base(n))))
    z1_terms, pp_terms = [], []
    routed, chunk, npp = cd["num_routed_wires"], cd["quotient_degree_factor"], cd["num_partial_products"]
    s_ids = [mul(x, base(k)) for k in cd["k_is"]]
    for i in range(cd["num_challenges"]):
        z_x, z_gx = op["plonk_zs"][i], op["plonk_zs_next"][i]
        z1_terms.append(sub(mul(l0, z_x), l0))
        beta, gamma = base(betas[i]), base(gammas[i])
        nums = [add(mul(beta, s_ids[j]), add(wires[j], gamma)) for j in range(routed)]
        dens = [add(mul(beta, op["plonk_sigmas"][j]), add(wires[j], gamma)) for j in range(routed)]
        accs = [z_x] + list(op["partial_products"][i * npp:(i + 1) * npp]) + [z_gx]
        for ch in range(0, routed, chunk):
            np_, dp = E1, E1
            for j in range(ch, min(ch + chunk, routed)):
                np_, dp = mul(np_, nums[j]), mul(dp, dens[j])
            prev, nxt = accs[ch // chunk], accs[ch // chunk + 1]
            pp_terms.append(sub(mul(prev, np_), mul(nxt, dp)))
    terms = z1_terms + pp_terms + allc
    return [reduce_with_powers(terms, base(a)) for a in alphas]


def verify(orc, cd, proof):
    """plonk_verifier_chip: challenges, vanishing identity at zeta, FRI.  Raises VerifyError."""
    nch = cd["num_challenges"]
    op = {k: [ext(v) for v in vals] for k, vals in proof["openings"].items()}
    fri = proof["opening_proof"]
    pi_hash = [int(v) for v in orc.hash_no_pad(np.asarray(proof["public_inputs"], dtype=np.uint64))]
    # ---- get_challenges --------------------------------------------------------------------------------
    hasher = cd.get("hasher", 0)          # GenericConfig::Hasher: Merkle trees + transcript (0 Poseidon, 1 the reference's Bn254PoseidonHash)
    tr = Transcript(orc, hasher)
    tr.observe(cd["circuit_digest"])
    tr.observe(pi_hash)
    tr.observe(proof["wires_cap"])
    betas, gammas = tr.squeeze(nch), tr.squeeze(nch)
    tr.observe(proof["plonk_zs_partial_products_cap"])
    alphas = tr.squeeze(nch)
    tr.observe(proof["quotient_polys_cap"])
    zeta = tuple(tr.squeeze(2))
    zeta_batch = op["constants"] + op["plonk_sigmas"] + op["wires"] + op["plonk_zs"] + op["partial_products"] + op["quotient_polys"]
    next_batch = op["plonk_zs_next"]
    for v in zeta_batch + next_batch:
        tr.observe(list(v))
    fri_alpha = tuple(tr.squeeze(2))
    fri_betas = []
    for cap in fri["commit_phase_merkle_caps"]:
        tr.observe(cap)
        fri_betas.append(tuple(tr.squeeze(2)))
    tr.observe(fri["final_poly"])
    tr.observe([fri["pow_witness"]])
    pow_response = tr.squeeze(1)[0]
    query_indices = tr.squeeze(cd["num_query_rounds"])
    # ---- vanishing identity ------------------------------------------------------------------------------
    n = 1 << cd["degree_bits"]
    zeta_pow_n = ext_pow(zeta, n)
    van = eval_vanishing_poly(cd, zeta, zeta_pow_n, op, pi_hash, betas, gammas, alphas)
    z_h = sub(zeta_pow_n, E1)
    qdf = cd["quotient_degree_factor"]
    for i in range(nch):
        chunk = op["quotient_polys"][i * qdf:(i + 1) * qdf]
        if mul(z_h, reduce_with_powers(chunk, zeta_pow_n)) != van[i]:
            raise VerifyError("quotient identity fails for challenge %d" % i)
    # ---- FRI ---------------------------------------------------------------------------------------------------
    if cd["pow_bits"] and pow_response >> (64 - cd["pow_bits"]):
        raise VerifyError("proof of work")
    g = pm.root_of_unity(cd["degree_bits"])
    zeta_next = (zeta[0] * g % P, zeta[1] * g % P)
    widths = [cd["num_selectors"] + cd["num_constants"] + cd["num_routed_wires"], cd["num_wires"],
              nch * (1 + cd["num_partial_products"]), nch * qdf]
    blinding = [False, True, True, True]
    all_polys = [(o, i) for o in range(4) for i in range(widths[o])]
    batches = [(zeta, all_polys, zeta_batch), (zeta_next, [(2, i) for i in range(nch)], next_batch)]
    reduced_openings = [reduce_with_powers(vals, fri_alpha) for _, _, vals in batches]
    caps = [cd["constants_sigmas_cap"], proof["wires_cap"], proof["plonk_zs_partial_products_cap"], proof["quotient_polys_cap"]]
    lde_bits = cd["degree_bits"] + cd["rate_bits"]
    N = 1 << lde_bits
    omega = pm.root_of_unity(lde_bits)
    if len(fri["query_round_proofs"]) != cd["num_query_rounds"] or len(fri["commit_phase_merkle_caps"]) != len(cd["arity_bits"]):
        raise VerifyError("proof shape")
    for q, rnd in zip(query_indices, fri["query_round_proofs"]):
        x_index = q % N
        if rnd["index"] != x_index:
            raise VerifyError("query index mismatch")
        for o, (leaf, sib) in enumerate(rnd["initial_trees"]):
            want = widths[o] + (4 if (cd["hiding"] and blinding[o]) else 0)
            if len(leaf) != want or not orc.merkle_verify(leaf, x_index, sib, caps[o], cd["cap_height"], hasher):
                raise VerifyError("initial tree %d opening" % o)
        x = 7 * pow(omega, pm.bitrev(x_index, lde_bits), P) % P
        # batch_initial_polynomials (fri_chip.rs:112-149)
        total = E0
        for (point, polys, _), red in zip(batches, reduced_openings):
            evals = [base(rnd["initial_trees"][o][0][i]) for o, i in polys]
            num = sub(reduce_with_powers(evals, fri_alpha), red)
            den = sub(base(x), point)
            total = add(mul(total, ext_pow(fri_alpha, len(evals))), mul(num, pm.ext_inv(den)))
        prev, idx, xx = total, x_index, x
        for i, arity_bits in enumerate(cd["arity_bits"]):
            assert arity_bits == 1
            evals_flat, sib = rnd["steps"][i]
            evals = [ext(evals_flat[0:2]), ext(evals_flat[2:4])]
            within, coset_index = idx & 1, idx >> 1
            if evals[within] != prev:
                raise VerifyError("fold consistency at layer %d" % i)
            # next_eval (fri_chip.rs:168-226): points x*g^i in bit-reversed order; arity 2 -> (x0, -x0)
            start = xx if within == 0 else (P - xx) % P
            a0, b0 = base(start), base((P - start) % P)
            a1, b1 = evals[0], evals[1]
            numer = mul(sub(fri_betas[i], a0), sub(b1, a1))
            prev = add(a1, mul(numer, pm.ext_inv(sub(b0, a0))))
            if not orc.merkle_verify(np.asarray(evals_flat, dtype=np.uint64), coset_index, sib, fri["commit_phase_merkle_caps"][i], cd["cap_height"], hasher):
                raise VerifyError("layer %d Merkle opening" % i)
            xx = xx * xx % P
            idx = coset_index
        final = reduce_with_powers([ext(c) for c in fri["final_poly"]], base(xx))
        if final != prev:
            raise VerifyError("final polynomial")
    return dict(betas=betas, gammas=gammas, alphas=alphas, zeta=zeta, fri_alpha=fri_alpha, query_indices=query_indices)
