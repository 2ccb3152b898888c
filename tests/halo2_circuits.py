"""Test circuits for the Halo2 prover beside the reference's chip shape (stark-verifier_amd/halo2_chips.py): built on purpose
from other ingredients, so that prover, oracle and verifier restatement meet on more than one family of descriptors."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import halo2_model as hm  # noqa: E402

h2 = importlib.import_module("stark-verifier_amd.halo2")
ch = importlib.import_module("stark-verifier_amd.halo2_chips")


def plonk_with_tuple_lookup(k, tb, seed=11):
    """A second circuit family, unlike the reference's chips on purpose: vanilla PLONK arithmetic (q_a a + q_b b + q_m a b + q_c + q_o c), a gate with
    a rotation (q_n (a(wX) - a - b)), a TWO-column lookup whose inputs are products (q_l a, q_l c) into the table (t, 3 t + 1), and a permutation
    over three advice columns and an instance column (degree 5, two permutation sets)."""
    cs = h2.ConstraintSystem()
    a, b, c = cs.advice_column(), cs.advice_column(), cs.advice_column()
    qa, qb, qm, qc, qo, qn, ql = (cs.fixed_column() for _ in range(7))
    t0, t1 = cs.lookup_table_column(), cs.lookup_table_column()
    pub = cs.instance_column()
    for col in (a, b, c, pub):
        cs.enable_equality(col)
    A, B, Cc = cs.query_advice(a), cs.query_advice(b), cs.query_advice(c)
    cs.create_gate("arith", [cs.query_fixed(qa) * A + cs.query_fixed(qb) * B + cs.query_fixed(qm) * A * B + cs.query_fixed(qc) + cs.query_fixed(qo) * Cc])
    cs.create_gate("next", [cs.query_fixed(qn) * (cs.query_advice(a, 1) - A - B)])
    cs.lookup("pair", [(cs.query_fixed(ql) * A, t0), (cs.query_fixed(ql) * Cc, t1)])
    w = ch.Witness(cs, k)
    rng = np.random.default_rng(seed)
    R = hm.R
    T = 1 << tb
    m = w.usable - 1
    av = [int(x) for x in rng.integers(1, 1 << 60, m + 1)]
    bv = [int(x) for x in rng.integers(0, 1 << 60, m + 1)]
    fx = {n: [0] * w.n for n in ("qa", "qb", "qm", "qc", "qo", "qn", "ql")}
    cv = [0] * (m + 1)
    for i in range(m):
        kind = i % 4
        fx["qo"][i] = R - 1
        if kind == 0:                       # lookup row: c = 3 a + 1 with a in the table
            av[i] = int(rng.integers(1, T))
            fx["qa"][i], fx["qc"][i], fx["ql"][i] = 3, 1, 1
            cv[i] = 3 * av[i] + 1
        elif kind == 1:                     # a(next) = a + b
            fx["qa"][i], fx["qm"][i], fx["qn"][i] = 1, 1, 1
            cv[i] = (av[i] + av[i] * bv[i]) % R
            av[i + 1] = (av[i] + bv[i]) % R
            if (i + 1) % 4 == 0 and i + 1 < m:      # the next row is a lookup row: keep its a inside the table instead
                fx["qn"][i] = 0
                av[i + 1] = int(rng.integers(1, T))
        else:
            fx["qa"][i], fx["qb"][i], fx["qm"][i], fx["qc"][i] = 5, R - 2, 7, 12345
            cv[i] = (5 * av[i] - 2 * bv[i] + 7 * av[i] * bv[i] + 12345) % R
    # lookup rows whose a was overwritten by a preceding "next" row have to be recomputed
    for i in range(0, m, 4):
        cv[i] = 3 * av[i] + 1
    for i in range(m):
        w.set_int(w.advice, a.index, i, av[i]); w.set_int(w.advice, b.index, i, bv[i]); w.set_int(w.advice, c.index, i, cv[i])
    for name, col in zip(("qa", "qb", "qm", "qc", "qo", "qn", "ql"), (qa, qb, qm, qc, qo, qn, ql)):
        for i in range(m):
            if fx[name][i]:
                w.set_int(w.fixed, col.index, i, fx[name][i])
    for i in range(w.n):                    # (0, 0) for the rows that look nothing up, then (t, 3 t + 1)
        t = i if 1 <= i < T else 0
        w.set_int(w.fixed, t0.index, i, t)
        w.set_int(w.fixed, t1.index, i, 3 * t + 1 if t else 0)
    # copies: b of a plain row takes the c of the row before it; the first a and one result are public
    for i in range(2, m, 4):
        w.set_int(w.advice, b.index, i + 1, cv[i])
        bv[i + 1] = cv[i]
        if (i + 1) % 4 == 3:
            cv[i + 1] = (5 * av[i + 1] - 2 * bv[i + 1] + 7 * av[i + 1] * bv[i + 1] + 12345) % R
            w.set_int(w.advice, c.index, i + 1, cv[i + 1])
        w.assembly.copy(c, i, b, i + 1)
    w.instance[pub.index] = [av[0], cv[3]]
    w.assembly.copy(a, 0, pub, 0)
    w.assembly.copy(c, 3, pub, 1)
    return cs, w
