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


def points_to_words(pts):
    """[(x, y) | None] -> [len][8] uint64 (x, y little-endian words; the identity = zeros): the layout of gl355_plonk_pk_commitments"""
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is not None:
            for j in range(4):
                out[i, j] = (p[0] >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
                out[i, 4 + j] = (p[1] >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def oracle_vk_digest(cs, k, pk):
    """the transcript's initial scalar recomputed on the checker's side: the oracle's own commitments and the oracle's own Keccak-256"""
    return h2.vk_digest(cs, k, points_to_words(pk.fixed_commitments), points_to_words(pk.sigma_commitments), hm.keccak256)


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


def random_circuit(k, seed):
    """A satisfiable circuit drawn from `seed`: 2-4 free advice columns and 1-2 free fixed columns of random values, an optional instance
    column; 1-3 gates q_g (E_g - d_g) where E_g is a random expression (sums, products, scalings, negations, constants; advice at rotations
    -2 .. 2, fixed at -1 .. 1, instance at 0 .. 1; degree <= 3) and d_g a result column assigned row by row; 0-2 lookups of 1-2 columns whose
    inputs are q_l F_j (F_j random, degree <= 2) and whose table columns are filled with exactly the tuples the rows need (one of them read
    through a scaled table expression); copy constraints among advice cells, into the instance column and from a fixed column.
    -> (cs, witness)"""
    import random
    rng = random.Random(seed)
    R = hm.R
    cs = h2.ConstraintSystem()
    n = 1 << k
    free = [cs.advice_column() for _ in range(rng.randint(2, 4))]
    ffix = [cs.fixed_column() for _ in range(rng.randint(1, 2))]
    inst = cs.instance_column() if rng.random() < 0.7 else None
    n_inst = rng.randint(1, 4) if inst is not None else 0

    def leaf():
        r = rng.random()
        if r < 0.6:
            return cs.query_advice(rng.choice(free), rng.randint(-2, 2))
        if r < 0.8:
            return cs.query_fixed(rng.choice(ffix), rng.randint(-1, 1))
        if r < 0.9 and inst is not None:
            return cs.query_instance(inst, rng.randint(0, 1))
        return h2.Expression.constant(rng.randrange(R))

    def expr(max_deg, depth=0):
        if max_deg <= 1 or depth >= 3 or rng.random() < 0.25:
            return leaf()
        r = rng.random()
        if r < 0.4:
            return expr(max_deg, depth + 1) + expr(max_deg, depth + 1)
        if r < 0.5:
            return expr(max_deg, depth + 1) - expr(max_deg, depth + 1)
        if r < 0.8:
            da = rng.randint(1, max_deg - 1)
            return expr(da, depth + 1) * expr(max_deg - da, depth + 1)
        if r < 0.9:
            return expr(max_deg, depth + 1) * rng.randrange(1, R)
        return -expr(max_deg, depth + 1)

    gates = []
    for g in range(rng.randint(1, 3)):
        d, q, e = cs.advice_column(), cs.fixed_column(), expr(3)
        cs.create_gate("g%d" % g, [cs.query_fixed(q) * (e - cs.query_advice(d))])
        gates.append((d, q, e))
    lookups = []
    for li in range(rng.randint(0, 2)):
        ql = cs.fixed_column()
        fs = [expr(2) for _ in range(rng.randint(1, 2))]
        tabs = [cs.lookup_table_column() for _ in fs]
        scale = rng.randrange(2, 1000) if rng.random() < 0.5 else None           # the first table column read through a scaled expression
        tab_exprs = [(cs.query_fixed(t) * scale if (j == 0 and scale) else cs.query_fixed(t)) for j, t in enumerate(tabs)]
        cs.lookup("l%d" % li, [(cs.query_fixed(ql) * f, te) for f, te in zip(fs, tab_exprs)])
        lookups.append((ql, fs, tabs, scale))
    eq_adv = rng.sample(free, rng.randint(1, len(free)))
    for c in eq_adv:
        cs.enable_equality(c)
    if inst is not None:
        cs.enable_equality(inst)
    const_col = ffix[0] if rng.random() < 0.5 else None
    if const_col is not None:
        cs.enable_equality(const_col)
    w = ch.Witness(cs, k)
    u = w.usable
    A = [[rng.randrange(R) for _ in range(u)] + [0] * (n - u) for _ in range(cs.num_advice)]
    F = [[0] * n for _ in range(cs.num_fixed)]
    for c in ffix:
        F[c.index] = [rng.randrange(R) for _ in range(u)] + [0] * (n - u)
    inst_vals = [rng.randrange(R) for _ in range(n_inst)]
    # copy constraints first (they fix values), among rows that exist
    for _ in range(rng.randint(1, 6)):
        ca, cb = rng.choice(eq_adv), rng.choice(eq_adv)
        ra, rb = rng.randrange(u), rng.randrange(u)
        A[cb.index][rb] = A[ca.index][ra]
        w.assembly.copy(ca, ra, cb, rb)
    cells_fixed = set()
    if inst is not None:
        for j in range(rng.randint(1, n_inst)):
            ca, ra = rng.choice(eq_adv), rng.randrange(u)
            inst_vals[j] = A[ca.index][ra]
            w.assembly.copy(ca, ra, inst, j)
            cells_fixed.add((ca.index, ra))
    if const_col is not None:
        for _ in range(2):
            ca, ra, rf = rng.choice(eq_adv), rng.randrange(u), rng.randrange(u)
            if (ca.index, ra) in cells_fixed:
                continue
            A[ca.index][ra] = F[const_col.index][rf]
            w.assembly.copy(const_col, rf, ca, ra)
    # the cycles may have chained cells: settle every cycle on one value (the representative's)
    perm_cols = cs.permutation
    for ci, col in enumerate(perm_cols):
        for r in range(n):
            rep_c, rep_r = (int(v) for v in w.assembly.aux[ci, r])
            if (rep_c, rep_r) == (ci, r):
                continue
            src = perm_cols[rep_c]
            val = A[src.index][rep_r] if src.kind == h2.ADVICE else (F[src.index][rep_r] if src.kind == h2.FIXED else (inst_vals[rep_r] if rep_r < n_inst else 0))
            if col.kind == h2.ADVICE:
                A[col.index][r] = val
            elif col.kind == h2.INSTANCE:
                inst_vals[r] = val
            else:
                assert F[col.index][r] == val or src.kind != h2.FIXED
                F[col.index][r] = val

    def at(row):
        def q(kind, qi):
            col, rot = cs.queries[kind][qi]
            rr = (row + rot) % n
            if kind == h2.ADVICE:
                return A[col][rr]
            if kind == h2.FIXED:
                return F[col][rr]
            return inst_vals[rr] if rr < n_inst else 0
        return q
    lo, hi = 2, u - 2
    for d, q, e in gates:
        for r in range(lo, hi):
            if rng.random() < 0.7:
                F[q.index][r] = 1
                A[d.index][r] = e.evaluate(at(r))
    for ql, fs, tabs, scale in lookups:
        rows = [r for r in range(lo, hi) if rng.random() < 0.4]
        for r in rows:
            F[ql.index][r] = 1
        tuples = [tuple([0] * len(fs))] + [tuple(f.evaluate(at(r)) for f in fs) for r in rows]
        inv = pow(scale, -1, R) if scale else 1
        for i in range(n):
            t = tuples[i] if i < len(tuples) else tuples[rng.randrange(len(tuples))]
            for j, tc in enumerate(tabs):
                F[tc.index][i] = t[j] * (inv if j == 0 else 1) % R
    for c in range(cs.num_advice):
        for r in range(n):
            if A[c][r]:
                w.set_int(w.advice, c, r, A[c][r])
    for c in range(cs.num_fixed):
        for r in range(n):
            if F[c][r]:
                w.set_int(w.fixed, c, r, F[c][r])
    if inst is not None:
        w.instance[inst.index] = inst_vals
    return cs, w
