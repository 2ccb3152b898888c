"""Recursive proof verification in-circuit and proof aggregation -- host mirror of
src/plonky2_semaphore/recursion.rs (aggregate_signals :25-185, aggregate :187-247) and wrapper.rs:35-56,
which call plonky2's `builder.verify_proof::<InnerC>()`.

verify_proof() below emits, through gadgets.GadgetBuilder, exactly the checks the reference's own
verifier performs on a proof (chip/plonk/plonk_verifier_chip.rs:55-242, chip/plonk/vanishing_poly.rs,
chip/plonk/gates/*.rs, chip/fri_chip.rs, chip/merkle_proof_chip.rs) -- the same equations
tests/plonk_verifier.py evaluates over big integers, here as gates over targets.  The resulting
circuit is proved by the same GPU pipeline (gl355_prove_sparse) as any other circuit.
"""
import ctypes as C

import numpy as np

from ._lib import (GATE_ARITHMETIC, GATE_ARITHMETIC_EXT, GATE_BASE_SUM, GATE_CONSTANT, GATE_NOOP, GATE_POSEIDON,
                   GATE_POSEIDON_MDS, GATE_PUBLIC_INPUT, GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT)
from .gadgets import CIRC, GadgetBuilder, T
from .plonk import CircuitConfig, P, _ptr, _u64, derive_key, parse_proof_tagged, prove_sparse

UNUSED_SELECTOR = 0xFFFFFFFF
_RC = None


def _round_constants():
    global _RC
    if _RC is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "poseidon_goldilocks_round_constants.txt")
        vals = [int(l, 16) for l in open(path).read().split("\n") if l and not l.startswith("#")]
        _RC = [vals[12 * r:12 * (r + 1)] for r in range(30)]
    return _RC


class Challenger:
    """in-circuit duplex sponge (hasher_chip.rs:48-89): buffered overwrite absorb, squeeze from the end of the rate"""

    def __init__(self, b):
        self.b = b
        self.state = [b.zero()] * 12
        self.inp, self.out = [], []

    def observe(self, targets):
        for t in targets:
            self.out = []
            self.inp.append(t)
            if len(self.inp) == 8:
                self._duplex()

    def observe_ext(self, e):
        self.observe([e[0], e[1]])

    def _duplex(self):
        st = list(self.inp) + self.state[len(self.inp):]
        self.inp = []
        self.state = self.b.permute_swapped(st)
        self.out = list(self.state[:8])

    def squeeze(self, n=1):
        res = []
        for _ in range(n):
            if self.inp or not self.out:
                self._duplex()
            res.append(self.out.pop())
        return res

    def squeeze_ext(self):
        v = self.squeeze(2)
        return (v[0], v[1])


# ---- gate constraint evaluators over extension targets (chip/plonk/gates/*.rs) -------------------------------
def _sbox(b, x):
    x2 = b.ext_mul(x, x)
    x4 = b.ext_mul(x2, x2)
    return b.ext_mul(b.ext_mul(x, x2), x4)


def eval_poseidon(b, w):
    rc = _round_constants()
    c = []
    swap = w[24]
    c.append(b.ext_mul_sub(swap, swap, swap))
    st = [None] * 12
    for i in range(4):
        lhs, rhs, delta = w[i], w[i + 4], w[25 + i]
        c.append(b.ext_mul_sub(swap, b.ext_sub(rhs, lhs), delta))
        st[i] = b.ext_add(lhs, delta)
        st[i + 4] = b.ext_sub(rhs, delta)
    for i in range(8, 12):
        st[i] = w[i]

    def add_rc(state, rnd):
        return [b.ext_add(x, b.constant_ext((rc[rnd][i], 0))) for i, x in enumerate(state)]
    for r in range(4):
        st = add_rc(st, r)
        if r != 0:
            for i in range(12):
                sin = w[29 + 12 * (r - 1) + i]
                c.append(b.ext_sub(st[i], sin))
                st[i] = sin
        st = b.mds_ext([_sbox(b, x) for x in st])
    # partial rounds, dense form: the S-box input of round r is state[0] + RC[4+r][0] in either formulation, so
    # these are the same constraint polynomials as the reference's sparse evaluation (poseidon.rs:652-673)
    for r in range(22):
        st = add_rc(st, 4 + r)
        sin = w[65 + r]
        c.append(b.ext_sub(st[0], sin))
        st[0] = _sbox(b, sin)
        st = b.mds_ext(st)
    for r in range(4):
        st = add_rc(st, 26 + r)
        for i in range(12):
            sin = w[87 + 12 * r + i]
            c.append(b.ext_sub(st[i], sin))
            st[i] = sin
        st = b.mds_ext([_sbox(b, x) for x in st])
    for i in range(12):
        c.append(b.ext_sub(st[i], w[12 + i]))
    assert len(c) == 123
    return c


def _alg_mul(b, x, y):
    """extension-algebra product of pairs of ext targets (goldilocks_extension_algebra_chip.rs:112-146)"""
    w7 = b.constant_ext((7, 0))
    c0 = b.ext_mul_add(b.ext_mul(x[1], y[1]), w7, b.ext_mul(x[0], y[0]))
    c1 = b.ext_mul_add(x[0], y[1], b.ext_mul(x[1], y[0]))
    return (c0, c1)


def eval_gate(b, gate, consts, w, pi_hash):
    t, p = gate
    if t == GATE_NOOP:
        return []
    if t == GATE_CONSTANT:
        return [b.ext_sub(consts[i], w[i]) for i in range(p)]
    if t == GATE_PUBLIC_INPUT:
        return [b.ext_sub(w[i], b.ext_from_base(pi_hash[i])) for i in range(4)]
    if t == GATE_BASE_SUM:
        limbs = w[1:1 + p]
        two = b.constant_ext((2, 0))
        acc = b.ext_zero()
        for l in reversed(limbs):
            acc = b.ext_mul_add(acc, two, l)
        return [b.ext_sub(acc, w[0])] + [b.ext_mul_sub(l, l, l) for l in limbs]
    if t == GATE_ARITHMETIC:
        return [b.ext_sub(w[4 * i + 3], b.ext_mul_add(b.ext_mul(w[4 * i], w[4 * i + 1]), consts[0], b.ext_mul(w[4 * i + 2], consts[1])))
                for i in range(p)]
    if t == GATE_POSEIDON:
        return eval_poseidon(b, w)

    def alg(j):
        return (w[j], w[j + 1])

    def asub(x, y):
        return (b.ext_sub(x[0], y[0]), b.ext_sub(x[1], y[1]))

    def aadd(x, y):
        return (b.ext_add(x[0], y[0]), b.ext_add(x[1], y[1]))

    def ascal(s, x):
        return (b.ext_mul(s, x[0]), b.ext_mul(s, x[1]))
    if t == GATE_ARITHMETIC_EXT:
        out = []
        for i in range(p):
            comp = aadd(ascal(consts[0], _alg_mul(b, alg(8 * i), alg(8 * i + 2))), ascal(consts[1], alg(8 * i + 4)))
            out += list(asub(alg(8 * i + 6), comp))
        return out
    if t == GATE_POSEIDON_MDS:
        out = []
        cc = [b.constant_ext((c, 0)) for c in CIRC]
        c8 = b.constant_ext((8, 0))
        for r in range(12):
            acc = (b.ext_zero(), b.ext_zero())
            for i in range(12):
                x = alg(2 * ((i + r) % 12))
                acc = (b.ext_mul_add(cc[i], x[0], acc[0]), b.ext_mul_add(cc[i], x[1], acc[1]))
            if r == 0:
                x = alg(0)
                acc = (b.ext_mul_add(c8, x[0], acc[0]), b.ext_mul_add(c8, x[1], acc[1]))
            out += list(asub(alg(2 * (12 + r)), acc))
        return out
    if t == GATE_RANDOM_ACCESS:
        bits, copies, extra = p & 0xFF, (p >> 8) & 0xFF, (p >> 16) & 0xFF
        vec = 1 << bits
        routed = (2 + vec) * copies + extra
        two = b.constant_ext((2, 0))
        out = []
        for c in range(copies):
            b0 = (2 + vec) * c
            bl = [w[routed + c * bits + i] for i in range(bits)]
            out += [b.ext_mul_sub(x, x, x) for x in bl]
            acc = b.ext_zero()
            for x in reversed(bl):
                acc = b.ext_mul_add(acc, two, x)
            out.append(b.ext_sub(acc, w[b0]))
            items = [w[b0 + 2 + i] for i in range(vec)]
            for x in bl:
                items = [b.ext_mul_add(x, b.ext_sub(items[2 * k + 1], items[2 * k]), items[2 * k]) for k in range(len(items) // 2)]
            out.append(b.ext_sub(items[0], w[b0 + 1]))
        out += [b.ext_sub(consts[i], w[(2 + vec) * copies + i]) for i in range(extra)]
        return out
    if t in (GATE_REDUCING, GATE_REDUCING_EXT):
        isext = t == GATE_REDUCING_EXT
        alpha, acc = alg(2), alg(4)
        start_accs = 6 + (2 * p if isext else p)
        out = []
        for i in range(p):
            coeff = alg(6 + 2 * i) if isext else (w[6 + i], b.ext_zero())
            acc_i = alg(0) if i == p - 1 else alg(start_accs + 2 * i)
            out += list(asub(aadd(_alg_mul(b, acc, alpha), coeff), acc_i))
            acc = acc_i
        return out
    raise NotImplementedError("gate %r has no in-circuit evaluator" % (gate,))


def eval_vanishing_poly(b, cd, x, x_pow_n, op, pi_hash, betas, gammas, alphas):
    """vanishing_poly.rs:18-153 over targets; returns one ext target per challenge"""
    n = 1 << cd["degree_bits"]
    consts, wires = op["constants"], op["wires"]
    allc = [None] * cd["num_gate_constraints"]
    for gi, gate in enumerate(cd["gates"]):
        sel = cd["selector_indices"][gi]
        lo, hi = cd["groups"][sel]
        f = consts[sel]
        filt = None
        for k in [k for k in range(lo, hi) if k != gi] + ([UNUSED_SELECTOR] if cd["num_selectors"] > 1 else []):
            term = b.ext_sub(b.constant_ext((k, 0)), f)
            filt = term if filt is None else b.ext_mul(filt, term)
        for k, c in enumerate(eval_gate(b, gate, consts[cd["num_selectors"]:], wires, pi_hash)):
            fc = c if filt is None else b.ext_mul(filt, c)
            allc[k] = fc if allc[k] is None else b.ext_add(allc[k], fc)
    allc = [c if c is not None else b.ext_zero() for c in allc]
    one = b.ext_one()
    # L0(x) = (x^n - 1) / (n (x - 1))
    l0 = b.ext_div(b.ext_sub(x_pow_n, one), b.arithmetic_ext(n, x, one, P - n, one))
    z1_terms, pp_terms = [], []
    routed, chunk, npp = cd["num_routed_wires"], cd["quotient_degree_factor"], cd["num_partial_products"]
    s_ids = [b.ext_mul(x, b.constant_ext((k, 0))) for k in cd["k_is"]]
    for i in range(cd["num_challenges"]):
        z_x, z_gx = op["plonk_zs"][i], op["plonk_zs_next"][i]
        z1_terms.append(b.ext_mul_sub(l0, z_x, l0))
        beta, gamma = b.ext_from_base(betas[i]), b.ext_from_base(gammas[i])
        nums, dens = [], []
        for j in range(routed):
            wg = b.ext_add(wires[j], gamma)
            nums.append(b.ext_mul_add(beta, s_ids[j], wg))
            dens.append(b.ext_mul_add(beta, op["plonk_sigmas"][j], wg))
        accs = [z_x] + list(op["partial_products"][i * npp:(i + 1) * npp]) + [z_gx]
        for ch in range(0, routed, chunk):
            np_, dp = nums[ch], dens[ch]
            for j in range(ch + 1, min(ch + chunk, routed)):
                np_, dp = b.ext_mul(np_, nums[j]), b.ext_mul(dp, dens[j])
            prev, nxt = accs[ch // chunk], accs[ch // chunk + 1]
            pp_terms.append(b.ext_mul_sub(prev, np_, b.ext_mul(nxt, dp)))
    terms = z1_terms + pp_terms + allc
    return [b.reduce_with_powers_ext(terms, b.ext_from_base(a)) for a in alphas]


def verify_merkle_proof(b, leaf, index_bits, cap_index, cap_targets, siblings):
    """merkle_proof_chip.rs:39-87: fold the path with swap = index bit, look the root up in the cap"""
    state = b.hash_or_noop(leaf)
    z = b.zero()
    for bit, sib in zip(index_bits, siblings):
        state = b.permute_swapped(list(state) + list(sib) + [z] * 4, swap=bit)[:4]
    n_cap = len(cap_targets)
    for i in range(4):
        items = [cap_targets[k][i] for k in range(n_cap)] + [z] * (16 - n_cap)
        root_i = b.random_access(cap_index, items) if n_cap > 1 else cap_targets[0][i]
        b.connect(root_i, state[i])


def verify_proof(b, cd, proof, register_pis=True):
    """builder.verify_proof: proof (dict as produced by plonk.parse_proof) becomes virtual targets; the inner
    circuit's verifier data (constants_sigmas cap, circuit digest) are constants.  Returns the inner public-input
    targets."""
    assert cd.get("hasher", 0) == 0, "the in-circuit verifier hashes with Poseidon-Goldilocks: the inner proof must use PoseidonHash"
    nch = cd["num_challenges"]
    tv = b.add_virtual_target

    def ext_list(vals):
        return [(tv(v[0]), tv(v[1])) for v in vals]

    def cap_targets(cap):
        return [[tv(x) for x in h] for h in cap]
    pis = [tv(v) for v in proof["public_inputs"]]
    if register_pis:
        b.register_public_inputs(pis)
    op = {k: ext_list(v) for k, v in proof["openings"].items()}
    wires_cap, zs_cap, q_cap = cap_targets(proof["wires_cap"]), cap_targets(proof["plonk_zs_partial_products_cap"]), cap_targets(proof["quotient_polys_cap"])
    cs_cap = [[b.constant(int(x)) for x in h] for h in cd["constants_sigmas_cap"]]
    fri = proof["opening_proof"]
    fri_caps = [cap_targets(c) for c in fri["commit_phase_merkle_caps"]]
    final_poly = ext_list(fri["final_poly"])
    pow_witness = tv(fri["pow_witness"])
    # ---- challenges (plonk_verifier_chip.rs:55-154) -----------------------------------------------------------
    pi_hash = b.hash_n_to_hash_no_pad(pis)
    ch = Challenger(b)
    ch.observe([b.constant(int(x)) for x in cd["circuit_digest"]])
    ch.observe(pi_hash)
    for h in wires_cap:
        ch.observe(h)
    betas, gammas = ch.squeeze(nch), ch.squeeze(nch)
    for h in zs_cap:
        ch.observe(h)
    alphas = ch.squeeze(nch)
    for h in q_cap:
        ch.observe(h)
    zeta = ch.squeeze_ext()
    zeta_batch = op["constants"] + op["plonk_sigmas"] + op["wires"] + op["plonk_zs"] + op["partial_products"] + op["quotient_polys"]
    next_batch = op["plonk_zs_next"]
    for e in zeta_batch + next_batch:
        ch.observe_ext(e)
    fri_alpha = ch.squeeze_ext()
    fri_betas = []
    for cap in fri_caps:
        for h in cap:
            ch.observe(h)
        fri_betas.append(ch.squeeze_ext())
    for e in final_poly:
        ch.observe_ext(e)
    ch.observe([pow_witness])
    pow_response = ch.squeeze(1)[0]
    query_challenges = ch.squeeze(cd["num_query_rounds"])
    # ---- vanishing identity (plonk_verifier_chip.rs:174-210) -----------------------------------------------------
    zeta_pow_n = b.ext_exp_pow2(zeta, cd["degree_bits"])
    van = eval_vanishing_poly(b, cd, zeta, zeta_pow_n, op, pi_hash, betas, gammas, alphas)
    z_h = b.ext_sub(zeta_pow_n, b.ext_one())
    qdf = cd["quotient_degree_factor"]
    for i in range(nch):
        chunk = op["quotient_polys"][i * qdf:(i + 1) * qdf]
        b.connect_ext(b.ext_mul(z_h, b.reduce_with_powers_ext(chunk, zeta_pow_n)), van[i])
    # ---- FRI (fri_chip.rs) -------------------------------------------------------------------------------------------
    resp_bits = b.split_le_64(pow_response)
    for bit in resp_bits[64 - cd["pow_bits"]:]:
        b.assert_zero(bit)
    lde_bits = cd["degree_bits"] + cd["rate_bits"]
    cap_h = cd["cap_height"]
    g = pow(7, (P - 1) >> cd["degree_bits"], P)
    zeta_next = b.ext_mul(zeta, b.constant_ext((g, 0)))
    widths = [cd["num_selectors"] + cd["num_constants"] + cd["num_routed_wires"], cd["num_wires"],
              nch * (1 + cd["num_partial_products"]), nch * qdf]
    red_open = [b.reduce_with_powers_ext(zeta_batch, fri_alpha), b.reduce_with_powers_ext(next_batch, fri_alpha)]
    alpha_pows = [b.ext_exp_const(fri_alpha, len(zeta_batch)), b.ext_exp_const(fri_alpha, len(next_batch))]
    caps = [cs_cap, wires_cap, zs_cap, q_cap]
    omega = pow(7, (P - 1) >> lde_bits, P)
    one, zero = b.one(), b.zero()
    for q, rnd in zip(query_challenges, fri["query_round_proofs"]):
        b.begin_segment()             # a query round reads the transcript / openings above and its own wires only
        bits = b.split_le_64(q)[:lde_bits]
        cap_index = b.le_sum(bits[lde_bits - cap_h:]) if cap_h else zero
        leaves = [[tv(v) for v in leaf] for leaf, _ in rnd["initial_trees"]]
        for o in range(4):
            sib = [[tv(v) for v in s] for s in rnd["initial_trees"][o][1]]
            verify_merkle_proof(b, leaves[o], bits[:lde_bits - cap_h], cap_index, caps[o], sib)
        # x = 7 * omega^bitrev(index): bit j of the index contributes omega^(2^(lde_bits-1-j))
        x = b.constant(7)
        for j, bit in enumerate(bits):
            f = pow(omega, 1 << (lde_bits - 1 - j), P)
            x = b.mul(x, b.arithmetic(f - 1, bit, one, 1, one))       # bit ? f : 1
        # batch_initial_polynomials (fri_chip.rs:112-149)
        all_evals = [leaves[o][i] for o in range(4) for i in range(widths[o])]
        z_evals = [leaves[2][i] for i in range(nch)]
        xe = b.ext_from_base(x)
        total = b.ext_zero()
        for evals, red, apow, point in ((all_evals, red_open[0], alpha_pows[0], zeta), (z_evals, red_open[1], alpha_pows[1], zeta_next)):
            num = b.ext_sub(b.reduce_with_powers_base(evals, fri_alpha), red)
            den = b.ext_sub(xe, point)
            total = b.ext_mul_add(total, apow, b.ext_div(num, den))
        prev = total
        neg_one = b.constant(P - 1)
        idx_bits = bits
        for l, arity_bits in enumerate(cd["arity_bits"]):
            ev_flat, sib_vals = rnd["steps"][l]
            ev = [tv(v) for v in ev_flat]
            e0, e1 = (ev[0], ev[1]), (ev[2], ev[3])
            within = idx_bits[0]
            coset_bits = idx_bits[1:]
            picked = (b.select(within, e1[0], e0[0]), b.select(within, e1[1], e0[1]))
            b.connect_ext(picked, prev)
            # next_eval (fri_chip.rs:168-226): a0 = x * (-1)^within, b0 = -a0
            a0 = b.mul(x, b.arithmetic(P - 2, within, one, 1, one))   # within ? -x : x
            a0e = b.ext_from_base(a0)
            numer = b.ext_mul(b.ext_sub(fri_betas[l], a0e), b.ext_sub(e1, e0))
            denom = b.ext_from_base(b.mul_const(P - 2, a0))            # b0 - a0 = -2 a0
            prev = b.ext_add(e0, b.ext_div(numer, denom))
            sib = [[tv(v) for v in s] for s in sib_vals]
            verify_merkle_proof(b, ev, coset_bits[:len(coset_bits) - cap_h], cap_index, fri_caps[l], sib)
            x = b.mul(x, x)
            idx_bits = coset_bits
        fin = b.reduce_with_powers_ext(final_poly, b.ext_from_base(x))
        b.connect_ext(fin, prev)
        b.end_segment()
    return pis


def wrap_public_inputs(b, inner_pis):
    """wrapper.rs:35-47: the outer circuit re-exposes every inner public input, in order"""
    for pis in inner_pis:
        b.register_public_inputs(pis)


def aggregate_public_inputs(b, inner_pis):
    """recursion.rs:105-165: merkle_root | nullifiers(0 then 1) | topics(0 then 1).  Each inner proof exposes
    root | its nullifiers | its topics.  (The reference leaves the outer root an unconstrained virtual hash;
    here it IS inner proof 0's root and inner proof 1's root is connected to it -- same values, same layout,
    strictly more constrained.)"""
    root = inner_pis[0][:4]
    for pis in inner_pis[1:]:
        for j in range(4):
            b.connect(pis[j], root[j])
    b.register_public_inputs(root)
    halves = [(len(pis) - 4) // 2 for pis in inner_pis]
    for pis, h in zip(inner_pis, halves):
        b.register_public_inputs(pis[4:4 + h])
    for pis, h in zip(inner_pis, halves):
        b.register_public_inputs(pis[4 + h:4 + 2 * h])


class RecursiveCircuit:
    """wrapper.rs:35-56 WrapperCircuit / recursion.rs:25-185 aggregate_signals: a circuit that verifies k inner
    proofs of one inner circuit.  The first proof set fixes the layout (gate rows, selectors, sigmas) AND records
    the witness tape; every later proof set is witnessed by gl355_witness_replay (C, about 10 ms) straight from
    the inner proofs' flat words and proven by gl355_prove_sparse.  `prove_python` keeps the eager Python gadget
    pass (tests cross-check the tape against it)."""

    def __init__(self, ctx, inner_common, k=1, config=None, public_inputs=wrap_public_inputs):
        self.ctx, self.cd, self.k, self.config = ctx, inner_common, k, config
        self.pi_layout = public_inputs
        self.data = None
        self.structure = None
        self.tape = None

    def _run(self, proofs):
        b = GadgetBuilder(self.config)
        inner_pis = [verify_proof(b, self.cd, p, register_pis=False) for p in proofs]
        self.pi_layout(b, inner_pis)
        pi_vals = b.finalize_public_inputs()
        return b, pi_vals

    # ---- layout + tape from the first proof set ---------------------------------------------------------------
    def build(self, flat_proofs, rng=None):
        """flat_proofs: k pairs (flat proof words, public inputs) of VALID inner proofs"""
        assert len(flat_proofs) == self.k and self.data is None
        tagged, off = [], 0
        for flat, pi in flat_proofs:
            tagged.append(parse_proof_tagged(self.cd, flat, pi, off))
            off += len(flat) + len(pi)
        b, _ = self._run(tagged)
        assert getattr(b, "untagged_inputs", 0) == 0
        self.structure = b.structure_hash()
        self.tape, self.row_idx, self.pi_pos = b.witness_tape()
        self.tape_layout = b.tape_layout
        self.data = b.cb.build(self.ctx, rng)
        self.n_inputs = off
        return self

    def native(self):
        """the circuit + its witness tape as a loaded artifact (plonk.NativeCircuit): per proof one C call,
        gl355_circuit_prove_tape(inner proofs' flat words | public inputs)"""
        if getattr(self, "_native", None) is None:
            from .plonk import NativeCircuit
            self._native = NativeCircuit(self.ctx, self.data.export_blob(self.row_idx, self.tape, self.pi_pos, self.n_inputs, self.tape_layout))
        return self._native

    def witness(self, flat_proofs):
        """(rows uint64[n_rows][num_wires], public inputs) by tape replay; raises on an invalid inner proof"""
        inputs = np.concatenate([np.concatenate([_u64(f), _u64(p)]) for f, p in flat_proofs])
        assert inputs.size == self.n_inputs
        nw = self.data.config.num_wires
        rows = np.empty((self.row_idx.size, nw), dtype=np.uint64)
        failed = C.c_uint64(0)
        rc = self.ctx.lib.gl355_witness_replay(_ptr(self.tape), self.tape.shape[0], _ptr(inputs), inputs.size, _ptr(rows), rows.size,
                                               nw, C.byref(failed))
        if rc != 0:
            raise AssertionError("witness generation failed (rc %d) at tape entry %d: %r" % (rc, failed.value, self.tape[failed.value] if failed.value < self.tape.shape[0] else None))
        return rows, rows.reshape(-1)[self.pi_pos]

    def prove_flat(self, flat_proofs, seed, rng=None, flat_only=True):
        """-> (flat outer proof, outer public inputs) [or the parsed proof dict]"""
        if self.data is None:
            self.build(flat_proofs, rng)
        rows, pis = self.witness(flat_proofs)
        out = prove_sparse(self.ctx, self.data, self.row_idx, rows, pis, seed, flat_only=flat_only)
        return (out, pis) if flat_only else out

    # ---- the eager Python pass (parsed proof dicts) ---------------------------------------------------------
    def prove(self, proofs, seed, rng=None, flat_only=False):
        assert len(proofs) == self.k
        b, pi_vals = self._run(proofs)
        sh = b.structure_hash()
        if self.data is None:
            self.data = b.cb.build(self.ctx, rng)
            self.structure = sh
        assert sh == self.structure, "recursive circuit layout depends on the proof values"
        idx, vals = b.sparse_witness()
        return prove_sparse(self.ctx, self.data, idx, vals, np.array(pi_vals, dtype=np.uint64), seed, flat_only=flat_only)


class WrapperCircuit(RecursiveCircuit):
    """wrapper.rs:20-56: the final wrap.  One inner (Poseidon-Goldilocks) proof is verified in a circuit that is itself proven
    under OuterC = Bn254PoseidonGoldilocksConfig (access_set.rs:48-49): Merkle trees, transcript and proof-of-work of the
    OUTER proof use the reference's BN254-Poseidon hasher (bn245_poseidon/plonky2_config.rs:57-104), with
    `standard_stark_verifier_config()`: cap_height 0, no zero-knowledge blinding, rate 8, 28 queries, 16 PoW bits.  This is
    the proof the Halo2 verifier circuit consumes (out of scope)."""

    def __init__(self, ctx, inner_common):
        from .api import HASH_BN254_POSEIDON
        cfg = CircuitConfig(cap_height=0, zero_knowledge=False, hasher=HASH_BN254_POSEIDON)
        super().__init__(ctx, inner_common, k=1, config=cfg, public_inputs=wrap_public_inputs)


class Aggregator:
    """recursion.rs:187-247 `aggregate`: pairwise aggregation of 2^L signals into one proof.  Level l has its own
    circuit (it verifies two proofs of level l-1; level 0 is the Semaphore circuit) built from the first pair it
    sees; all pairs of a level are independent proofs."""

    def __init__(self, ctx, signal_common, config=None):
        self.ctx, self.config = ctx, config
        self.levels = []            # RecursiveCircuit per level
        self.commons = [signal_common]

    def aggregate(self, signals, seed=None, rng=None, ctxs=None, start_level=0, key_domain=0):
        """signals: list of (flat proof, public inputs), power-of-two many, all of the level-0 circuit and the
        same Merkle root.  Returns (flat proof, public inputs, common data of the final circuit).
        seed: None (the default) lets the library draw a fresh 256-bit blinding key from the OS CSPRNG for every proof (the
        zero-knowledge setting of recursion.rs:32-48).  A seed (integer or 32 bytes) makes the run reproducible: the key of node j of
        tree level l is gl355_derive_key(seed, key_domain << 48 | l << 32 | j), so no two proofs of a tree -- or, with a distinct
        `key_domain` per rank, of a distributed tree -- share a blinding stream.
        ctxs: prover contexts of the same device; the nodes of a level are independent and are proven on them in parallel
        (the reference's `par_chunks_exact(2)`, recursion.rs:211-227), one host thread per context.
        start_level: the signals are proofs of tree level `start_level` already (continuing a tree whose lower part was aggregated
        elsewhere, e.g. on other GPUs: parallel.aggregate_distributed)."""
        import threading
        n = len(signals)
        assert n >= 2 and n & (n - 1) == 0
        ctxs = list(ctxs) if ctxs else [self.ctx]
        level = start_level
        assert level < len(self.commons), "the circuit of the incoming proofs is not known yet"
        while len(signals) > 1:
            if level == len(self.levels):
                self.levels.append(RecursiveCircuit(self.ctx, self.commons[level], k=2, config=self.config,
                                                    public_inputs=aggregate_public_inputs))
            rc = self.levels[level]
            if rc.data is None:
                rc.build(signals[0:2], rng)
            nat = rc.native()
            n_nodes = len(signals) // 2
            nxt, errors = [None] * n_nodes, []
            cur = signals

            # the nodes of a level are independent proofs of one circuit: every context proves its share in lock-step batches
            # (gl355_circuit_prove_tape_units: one launch per stage for up to `units` nodes)
            units = max(1, min(8, -(-n_nodes // len(ctxs))))

            def worker(t, cur=cur, nxt=nxt, nat=nat, n_nodes=n_nodes, units=units, lvl=level):
                try:
                    mine = list(range(t, n_nodes, len(ctxs)))
                    for b in range(0, len(mine), units):
                        js = mine[b:b + units]
                        inputs = np.stack([np.concatenate([np.concatenate([_u64(f), _u64(p)]) for f, p in cur[2 * j:2 * j + 2]]) for j in js])
                        keys = None if seed is None else [derive_key(seed, (int(key_domain) << 48) | (lvl << 32) | j) for j in js]
                        flats, pis = nat.prove_tape_units(ctxs[t], inputs, keys)
                        for k, j in enumerate(js):
                            nxt[j] = (flats[k], pis[k])
                except Exception as exc:
                    errors.append(exc)
            ths = [threading.Thread(target=worker, args=(t,)) for t in range(min(len(ctxs), n_nodes))]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            if errors:
                raise errors[0]
            if level + 1 == len(self.commons):
                self.commons.append(rc.data.common())
            signals = nxt
            level += 1
        return signals[0][0], signals[0][1], self.commons[level]

    # ---- the same tree through ONE native call (gl355_aggregate_units), and the level circuits as persisted artifacts --------------------
    def native_levels(self, n_levels):
        """the loaded artifacts (plonk.NativeCircuit) of the first n_levels level circuits"""
        assert n_levels <= len(self.levels) and all(rc.data is not None for rc in self.levels[:n_levels]), "build the level circuits first (aggregate once, or load())"
        return [rc.native() for rc in self.levels[:n_levels]]

    def aggregate_native(self, signals, seed=None, ctxs=None, key_domain=0, timed=False):
        """recursion.rs:187-247 by gl355_aggregate_units: every level's nodes in lock-step over the contexts, no Python between the
        proofs.  Same keys as `aggregate` -> the same bytes on a seeded run.  -> (flat proof, public inputs, common data[, level ms])"""
        from .plonk import key_bytes
        n = len(signals)
        n_levels = n.bit_length() - 1
        assert n >= 2 and n == 1 << n_levels
        nats = self.native_levels(n_levels) if getattr(self, "_loaded", None) is None else self._loaded[:n_levels]
        ctxs = list(ctxs) if ctxs else [self.ctx]
        lib = self.ctx.lib
        handles = (C.c_void_p * n_levels)(*[nat.h for nat in nats])
        cs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        proofs = np.ascontiguousarray(np.stack([_u64(f) for f, _ in signals]))
        pis = np.ascontiguousarray(np.stack([_u64(p) for _, p in signals]))
        out = np.empty(nats[-1].proof_words, dtype=np.uint64)
        opis = np.empty(nats[-1].n_public_inputs, dtype=np.uint64)
        ms = np.zeros(n_levels, dtype=np.float64)
        key = None if seed is None else key_bytes(seed)
        rc = lib.gl355_aggregate_units(cs, len(ctxs), handles, n_levels, _ptr(proofs), _ptr(pis), n, proofs.shape[1], pis.shape[1], key, int(key_domain),
                                       _ptr(out), out.size, _ptr(opis), opis.size, ms.ctypes.data)
        if rc != 0:
            from . import _lib
            raise _lib.Gl355Error(rc, "; ".join((lib.gl355_last_error(c.h) or b"").decode() for c in ctxs))
        res = (out, opis, self.commons[n_levels] if n_levels < len(self.commons) else None)
        return res + ([float(v) for v in ms],) if timed else res

    def save(self, directory):
        """the level circuits as artifacts (gl355_circuit_load format) + the common data of every level: a later process aggregates without
        building anything (the reference rebuilds every level circuit inside every aggregate_signals call, recursion.rs:25-185, 167)"""
        import os
        import pickle
        os.makedirs(directory, exist_ok=True)
        for l, rc in enumerate(self.levels):
            if rc.data is None:
                break
            np.save(os.path.join(directory, "level%d.npy" % l), rc.native().blob)
        with open(os.path.join(directory, "commons.pkl"), "wb") as f:
            pickle.dump(self.commons, f)

    @classmethod
    def load(cls, ctx, directory):
        """-> an Aggregator whose aggregate_native runs from the persisted artifacts (no gadget pass, no circuit build)"""
        import os
        import pickle
        from .plonk import NativeCircuit
        with open(os.path.join(directory, "commons.pkl"), "rb") as f:
            commons = pickle.load(f)
        agg = cls(ctx, commons[0])
        agg.commons = commons
        agg._loaded = []
        l = 0
        while os.path.exists(os.path.join(directory, "level%d.npy" % l)):
            agg._loaded.append(NativeCircuit(ctx, np.load(os.path.join(directory, "level%d.npy" % l))))
            l += 1
        return agg

