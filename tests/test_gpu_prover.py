"""GPU: a full Semaphore proof (make_signal, access_set.rs:61-104) produced by the HIP pipeline must pass
the restatement of the reference's verifier (tests/plonk_verifier.py); the constraint kernel (a10) is
additionally compared point-by-point with the big-integer evaluation of vanishing_poly.rs."""
import importlib

import numpy as np
import pytest

import plonk_verifier as pv
import pymodel as pm
from oracle_lib import P, rand_field

pytestmark = pytest.mark.gpu


def make_access_set(gl, ctx, log_members, seed):
    sem = importlib.import_module("stark-verifier_amd.semaphore")
    rng = np.random.default_rng(seed)
    sks = rand_field(rng, (1 << log_members, 4))
    keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))     # signal.rs:32-39
    return sem.AccessSet(ctx, keys), sks, rng


def test_gate_witness_and_host_hash(gl, ctx, orc):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rng = np.random.default_rng(0x501)
    x = rand_field(rng, 135)
    assert np.array_equal(plonk.host_hash_no_pad(x), orc.hash_no_pad(x))
    for swap in (0, 1):
        inp = rand_field(rng, 12)
        w = plonk.poseidon_gate_witness(inp, swap)
        st = inp.copy()
        if swap:
            st[:4], st[4:8] = inp[4:8].copy(), inp[:4].copy()
        assert np.array_equal(w[12:24], orc.permute(st))
        cons = pv.eval_poseidon(None, [pv.base(v) for v in w], None)
        assert all(c == (0, 0) for c in cons)
        w2 = w.copy()
        w2[70] ^= np.uint64(1)
        assert any(c != (0, 0) for c in pv.eval_poseidon(None, [pv.base(v) for v in w2], None))
    # challenger == oracle challenger
    ch, oc = plonk.Challenger(), orc.challenger()
    for chunk in (3, 8, 1, 13):
        e = rand_field(rng, chunk)
        ch.observe(e)
        orc.observe(oc, e)
        assert int(ch.squeeze(1)[0]) == orc.squeeze(oc)
    st, pos = ch.pow_state()
    ch.observe(rand_field(rng, 2))
    st, pos = ch.pow_state()
    assert pos == 2


@pytest.mark.parametrize("log_members", [4])
def test_semaphore_proof_verifies(gl, ctx, orc, log_members):
    aset, sks, rng = make_access_set(gl, ctx, log_members, 0x357)
    topic = rand_field(rng, 4)
    signal, data = aset.make_signal(sks[12], topic, 12, np.random.default_rng(0x358), check=True)
    cd = data.common()
    ch = pv.verify(orc, cd, signal.proof)
    assert len(ch["query_indices"]) == 28
    # public inputs are root | nullifier | topic (access_set.rs:33-41)
    pi = signal.proof["public_inputs"]
    assert np.array_equal(pi[:4], aset.tree.cap[0]) and np.array_equal(pi[8:], topic)
    assert np.array_equal(pi[4:8], orc.hash_no_pad(np.concatenate([sks[12], topic])))
    # tampering is rejected
    bad = dict(signal.proof)
    bad["public_inputs"] = pi.copy()
    bad["public_inputs"][9] ^= np.uint64(1)
    with pytest.raises(pv.VerifyError):
        pv.verify(orc, cd, bad)
    bad = dict(signal.proof)
    bad["openings"] = dict(signal.proof["openings"])
    w = bad["openings"]["wires"].copy()
    w[5][0] = (int(w[5][0]) + 1) % P
    bad["openings"]["wires"] = w
    with pytest.raises(pv.VerifyError):
        pv.verify(orc, cd, bad)
    # a second proof from the same circuit with a different member and fresh blinding also verifies
    signal2, _ = aset.make_signal(sks[3], topic, 3, np.random.default_rng(0x359))
    pv.verify(orc, cd, signal2.proof)
    assert not np.array_equal(signal2.proof["wires_cap"], signal.proof["wires_cap"])
    # the stage-by-stage Python sequencing over the individual entry points gives a valid proof too
    signal3, _ = aset.make_signal(sks[7], topic, 7, np.random.default_rng(0x35A), staged=True)
    pv.verify(orc, cd, signal3.proof)


def test_sparse_witness_path(gl, ctx, orc):
    """gl355_semaphore_witness + gl355_prove_sparse (device-side blinding) == the dense witness on the real rows, and
    the proof verifies."""
    aset, sks, rng = make_access_set(gl, ctx, 5, 0x35E)
    data, rows = aset.build(rng)
    topic = rand_field(rng, 4)
    dense, pi = aset.fill_semaphore_targets(data, rows, sks[9], topic, 9, np.random.default_rng(3))
    idx, vals, pi2 = aset.witness_rows(rows, sks[9], topic, 9)
    assert np.array_equal(pi, pi2)
    for k, r in enumerate(idx):
        assert np.array_equal(dense[:, r], vals[k]), r
    sig, _ = aset.make_signal_fast(sks[9], topic, 9, 77)
    pv.verify(orc, data.common(), sig.proof)
    assert np.array_equal(sig.nullifier[0], orc.hash_no_pad(np.concatenate([sks[9], topic])))


def test_prove_is_deterministic_in_witness_and_seed(gl, ctx):
    """same witness + same seed => byte-identical proof (salt is counter-based, PoW takes the smallest witness)."""
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x35D)
    data, rows = aset.build(rng)
    wires, pi = aset.fill_semaphore_targets(data, rows, sks[2], rand_field(rng, 4), 2, np.random.default_rng(5))
    a = plonk.prove(ctx, data, wires, pi, 42, flat_only=True)
    b = plonk.prove(ctx, data, wires, pi, 42, flat_only=True)
    c = plonk.prove(ctx, data, wires, pi, 43, flat_only=True)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_proof_bytes_do_not_depend_on_the_ntt_pass_structure(gl):
    """GL355_OPT_NTT_SINGLE_PASS_MAX_LOG = 12 (default: the 2^13-point LDEs of the Semaphore circuit in two passes, streaming column kernel +
    limb rows) and = 14 (one pass, radix-8 single-tile kernels) give the same proof, byte for byte"""
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    proofs = []
    for max_log in (12, 14):
        c = gl.Context(0)
        c.set_option(4, max_log)
        aset, sks, rng = make_access_set(gl, c, 3, 0x35E)
        data, rows = aset.build(rng)
        wires, pi = aset.fill_semaphore_targets(data, rows, sks[5], rand_field(rng, 4), 5, np.random.default_rng(6))
        assert data.degree_bits == 13
        proofs.append(plonk.prove(c, data, wires, pi, 77, flat_only=True))
        c.close()
    assert np.array_equal(proofs[0], proofs[1])


def test_wrong_witness_fails_quotient(gl, ctx, orc):
    """a witness that violates a gate constraint yields a 'quotient' of full degree: the proof must not verify."""
    sem = importlib.import_module("stark-verifier_amd.semaphore")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x35A)
    data, rows = aset.build(rng)
    wires, pi = aset.fill_semaphore_targets(data, rows, sks[1], rand_field(rng, 4), 1, rng)
    wires[40, rows["null"]] ^= np.uint64(1)        # break one S-box wire
    for proof in (plonk.prove(ctx, data, wires, pi, 1), plonk.prove_staged(ctx, data, wires, pi, np.random.default_rng(1))):
        with pytest.raises(pv.VerifyError):
            pv.verify(orc, data.common(), proof)


def test_quotient_kernel_pointwise(gl, ctx, orc):
    """a10: vanishing(x)/Z_H(x) from the HIP kernel == big-integer evaluation of vanishing_poly.rs at sample points."""
    import ctypes as C
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    api = importlib.import_module("stark-verifier_amd.api")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x35B)
    data, rows = aset.build(rng)
    cfg = data.config
    wires, pi = aset.fill_semaphore_targets(data, rows, sks[5], rand_field(rng, 4), 5, rng)
    n, N = 1 << data.degree_bits, 1 << (data.degree_bits + cfg.rate_bits)
    wb = gl.PolynomialBatch.from_values(ctx, wires, cfg.rate_bits, cfg.cap_height, salt=rand_field(rng, (4, N)))
    betas, gammas, alphas = rand_field(rng, 2), rand_field(rng, 2), rand_field(rng, 2)
    zs, pps = [], []
    for c in range(2):
        z, pp = ctx.zs_partial_products(wires[:80], data.sigmas, data.k_is, 8, int(betas[c]), int(gammas[c]))
        zs.append(z)
        pps.append(pp)
    zb = gl.PolynomialBatch.from_values(ctx, np.concatenate([np.stack(zs)] + pps), cfg.rate_bits, cfg.cap_height)
    pi_hash = plonk.host_hash_no_pad(pi)
    vals = np.empty((2, N), dtype=np.uint64)
    ctx.check(ctx.lib.gl355_quotient_values(ctx.h, C.byref(data.c_circuit), data.constants_sigmas.h, wb.h, zb.h, api._ptr(data.k_is),
                                            api._ptr(betas), api._ptr(gammas), api._ptr(alphas), api._ptr(pi_hash), api._ptr(vals)))
    cs_leaves, w_leaves, z_leaves = data.constants_sigmas.leaves(), wb.leaves(), zb.leaves()
    cd = data.common()
    bits = data.degree_bits + cfg.rate_bits
    omega = pm.root_of_unity(bits)
    n_const = data.num_selectors + cfg.num_constants
    for t in (0, 1, 77, N // 2 + 5, N - 1):
        i = pm.bitrev(t, bits)
        x = 7 * pow(omega, i, P) % P
        t_next = pm.bitrev((i + 8) % N, bits)
        op = dict(constants=[pv.base(v) for v in cs_leaves[t][:n_const]], plonk_sigmas=[pv.base(v) for v in cs_leaves[t][n_const:]],
                  wires=[pv.base(v) for v in w_leaves[t][:135]], plonk_zs=[pv.base(v) for v in z_leaves[t][:2]],
                  partial_products=[pv.base(v) for v in z_leaves[t][2:20]], plonk_zs_next=[pv.base(v) for v in z_leaves[t_next][:2]])
        xn = pow(x, n, P)
        van = pv.eval_vanishing_poly(cd, pv.base(x), pv.base(xn), op, [int(v) for v in pi_hash], [int(b) for b in betas],
                                     [int(g) for g in gammas], [int(a) for a in alphas])
        zh_inv = pow((xn - 1) % P, P - 2, P)
        for c in range(2):
            assert van[c][1] == 0
            assert int(vals[c][t]) == van[c][0] * zh_inv % P, (t, c)
    # and the interpolated quotient has degree < 8n by construction; its chunks recombine to the values
    q = np.empty((16, n), dtype=np.uint64)
    ctx.check(ctx.lib.gl355_quotient(ctx.h, C.byref(data.c_circuit), data.constants_sigmas.h, wb.h, zb.h, api._ptr(data.k_is),
                                     api._ptr(betas), api._ptr(gammas), api._ptr(alphas), api._ptr(pi_hash), api._ptr(q)))
    full = q.reshape(2, 8 * n)
    back = ctx.coset_fft(full)            # natural order values on 7<omega_N>
    assert np.array_equal(orc.reverse_index_bits(back.T.copy()).T, vals)
    wb.close()
    zb.close()


def test_gate_set_pointwise(gl, ctx):
    """a10, the whole gate set of gates/mod.rs:141-196 (+ BaseSum{20} of circuit.rs:42): random wires / constants /
    selectors, HIP kernel vs big-integer evaluation of the same formulas at sample points (the reference's
    gate_test.rs methodology: random inputs, compare two evaluators)."""
    import ctypes as C
    lib = importlib.import_module("stark-verifier_amd._lib")
    api = importlib.import_module("stark-verifier_amd.api")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rng = np.random.default_rng(0x35C)
    gates = [(lib.GATE_POSEIDON, 0), (lib.GATE_ARITHMETIC, 20), (lib.GATE_PUBLIC_INPUT, 0), (lib.GATE_NOOP, 0), (lib.GATE_CONSTANT, 2),
             (lib.GATE_BASE_SUM, 63), (lib.GATE_BASE_SUM, 4), (lib.GATE_BASE_SUM, 20), (lib.GATE_POSEIDON_MDS, 0),
             (lib.GATE_RANDOM_ACCESS, 1 | 20 << 8), (lib.GATE_RANDOM_ACCESS, 4 | 4 << 8 | 2 << 16), (lib.GATE_REDUCING_EXT, 32),
             (lib.GATE_REDUCING, 43), (lib.GATE_ARITHMETIC_EXT, 10), (lib.GATE_MUL_EXT, 13)]
    groups = [(0, 1), (1, 6), (6, 11), (11, 15)]
    sel_idx = [next(s for s, (lo, hi) in enumerate(groups) if lo <= g < hi) for g in range(len(gates))]
    degree_bits, rate_bits, nch = 4, 3, 2
    n, N = 1 << degree_bits, 1 << (degree_bits + rate_bits)
    n_sel, n_cst, routed, nw, npp = len(groups), 2, 80, 135, 9
    cs_vals = rand_field(rng, (n_sel + n_cst + routed, n))
    cs_vals[:n_sel] = rng.integers(0, len(gates), size=(n_sel, n)).astype(np.uint64)   # selector-like small values
    cs = gl.PolynomialBatch.from_values(ctx, cs_vals, rate_bits, 2)
    wb = gl.PolynomialBatch.from_values(ctx, rand_field(rng, (nw, n)), rate_bits, 2)
    zb = gl.PolynomialBatch.from_values(ctx, rand_field(rng, (nch * (1 + npp), n)), rate_bits, 2)
    cc = lib.Circuit()
    cc.degree_bits, cc.rate_bits, cc.num_wires, cc.num_routed_wires = degree_bits, rate_bits, nw, routed
    cc.num_constants, cc.num_selectors, cc.num_challenges, cc.max_degree = n_cst, n_sel, nch, 8
    cc.num_partial_products, cc.num_gates = npp, len(gates)
    for i, (t, p) in enumerate(gates):
        cc.gates[i].type, cc.gates[i].param, cc.gates[i].selector_index = t, p, sel_idx[i]
        cc.gates[i].group_start, cc.gates[i].group_end = groups[sel_idx[i]]
    k_is = np.array([pow(7, j, P) for j in range(routed)], dtype=np.uint64)
    betas, gammas, alphas, pi_hash = rand_field(rng, 2), rand_field(rng, 2), rand_field(rng, 2), rand_field(rng, 4)
    vals = np.empty((nch, N), dtype=np.uint64)
    ctx.check(ctx.lib.gl355_quotient_values(ctx.h, C.byref(cc), cs.h, wb.h, zb.h, api._ptr(k_is), api._ptr(betas), api._ptr(gammas),
                                            api._ptr(alphas), api._ptr(pi_hash), api._ptr(vals)))
    cd = dict(degree_bits=degree_bits, gates=gates, groups=groups, selector_indices=sel_idx, num_selectors=n_sel,
              num_constants=n_cst, num_wires=nw, num_routed_wires=routed, num_challenges=nch, quotient_degree_factor=8,
              num_partial_products=npp, num_gate_constraints=max(plonk._GATE_CONSTRAINTS[t](p) for t, p in gates),
              k_is=[int(k) for k in k_is])
    cs_l, w_l, z_l = cs.leaves(), wb.leaves(), zb.leaves()
    bits = degree_bits + rate_bits
    omega = pm.root_of_unity(bits)
    for t in (0, 3, 64, N - 1):
        i = pm.bitrev(t, bits)
        x = 7 * pow(omega, i, P) % P
        t_next = pm.bitrev((i + 8) % N, bits)
        op = dict(constants=[pv.base(v) for v in cs_l[t][:n_sel + n_cst]], plonk_sigmas=[pv.base(v) for v in cs_l[t][n_sel + n_cst:]],
                  wires=[pv.base(v) for v in w_l[t]], plonk_zs=[pv.base(v) for v in z_l[t][:nch]],
                  partial_products=[pv.base(v) for v in z_l[t][nch:]], plonk_zs_next=[pv.base(v) for v in z_l[t_next][:nch]])
        xn = pow(x, n, P)
        van = pv.eval_vanishing_poly(cd, pv.base(x), pv.base(xn), op, [int(v) for v in pi_hash], [int(b) for b in betas],
                                     [int(g) for g in gammas], [int(a) for a in alphas])
        zh_inv = pow((xn - 1) % P, P - 2, P)
        for c in range(nch):
            assert van[c][1] == 0 and int(vals[c][t]) == van[c][0] * zh_inv % P, (t, c)
    for o in (cs, wb, zb):
        o.close()
