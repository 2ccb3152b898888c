"""The reference's two Halo2 chips as circuit descriptions and witness assignment: the column / gate / lookup shape its
finalisation proof is made over.

  ArithmeticChipConfig.configure    chip/native_chip/arithmetic_chip.rs:44-160: a b c q r + 5 q-limbs + 4 r-limbs (14 advice), one fixed
                                    `constant`, four selectors, one 16-bit lookup table column, one instance column; gates "limb
                                    decomposition", "q = p - r", "base field constraint" (a b + c = q p + r) and the extension-field form; nine
                                    16-bit range lookups; equality on a b c r q instance constant
  PoseidonBn254ChipConfig.configure chip/native_chip/poseidon_bn254_chip.rs:27-123: five state columns, five round-constant columns, the
                                    partial-round and full-round gates (x^5, dense 5x5 MDS), equality on the state
  AllChipConfig.configure           chip/native_chip/all_chip.rs:22-30: both in one constraint system = `Verifier::configure`
                                    (verifier_circuit.rs:143-146)
and row-filling in the shape of ArithmeticChip::assign / assign_value (arithmetic_chip.rs:259-306) and PoseidonBn254Chip::apply_permute
(poseidon_bn254_chip.rs:203-233).  `synthetic_circuit` fills a 2^k-row instance with that shape for tests and benchmarks: the Halo2
verifier circuit itself (the layout of a plonky2 proof's verification over these chips) is OUT OF SCOPE."""
import os
import sys

import numpy as np

from . import halo2 as h2

GOLDILOCKS_MODULUS = ((1 << 32) - 1) * (1 << 32) + 1
Q_LIMBS, T, R_F, R_P = 5, 5, 8, 60
R = h2.R


class ArithmeticChipConfig:
    @staticmethod
    def configure(meta, table_bits=16, modulus=GOLDILOCKS_MODULUS):
        """table_bits / modulus: the reference's values by default; small-k tests shrink both together (a 2^16-row table needs k >= 17):
        limbs of `table_bits` bits, five for q and four for r, so modulus <= 2^(4 table_bits)"""
        c = ArithmeticChipConfig()
        c.table_bits, c.modulus = table_bits, modulus
        assert modulus <= 1 << (4 * table_bits)
        c.a, c.b, c.c, c.q, c.r = (meta.advice_column() for _ in range(5))
        c.q_limbs = [meta.advice_column() for _ in range(Q_LIMBS)]
        c.r_limbs = [meta.advice_column() for _ in range(4)]
        c.constant = meta.fixed_column()
        c.s_limb, c.s_range, c.s_base, c.s_ext = (meta.selector() for _ in range(4))
        c.table = meta.lookup_table_column()
        c.instance = meta.instance_column()
        for col in (c.a, c.b, c.c, c.r, c.q, c.instance):
            meta.enable_equality(col)
        meta.enable_constant(c.constant)
        cur, nxt = h2.Rotation.cur(), h2.Rotation.next()
        qa = meta.query_advice
        # "limb decomposition"
        s_limb = meta.query_selector(c.s_limb)
        q = qa(c.q, cur)
        q_acc = h2.Expression.constant(0)
        for i, l in enumerate(c.q_limbs):
            q_acc = q_acc + qa(l, cur) * (1 << (i * table_bits))
        r = qa(c.r, cur)
        r_acc = h2.Expression.constant(0)
        for i, l in enumerate(c.r_limbs):
            r_acc = r_acc + qa(l, cur) * (1 << (i * table_bits))
        meta.create_gate("limb decomposition", [s_limb * (q - q_acc), s_limb * (r - r_acc)])
        p = h2.Expression.constant(modulus)
        # "q = p - r"
        meta.create_gate("q = p - r", [meta.query_selector(c.s_range) * (qa(c.q, cur) - p + qa(c.r, cur))])
        # "base field constraint"
        s_base = meta.query_selector(c.s_base)
        meta.create_gate("base field constraint", [s_base * (qa(c.a, cur) * qa(c.b, cur) + qa(c.c, cur) - p * qa(c.q, cur) - qa(c.r, cur))])
        # "extension field contraint"
        s_ext = meta.query_selector(c.s_ext)
        ax, ay, bx, by = qa(c.a, cur), qa(c.a, nxt), qa(c.b, cur), qa(c.b, nxt)
        cx, cy, qx, qy, rx, ry = qa(c.c, cur), qa(c.c, nxt), qa(c.q, cur), qa(c.q, nxt), qa(c.r, cur), qa(c.r, nxt)
        left_x = ax * bx + h2.Expression.constant(7) * ay * by + cx
        left_y = ax * by + ay * bx + cy
        meta.create_gate("extension field contraint", [s_ext * (left_x - (p * qx + rx)), s_ext * (left_y - (p * qy + ry))])
        for l in c.q_limbs:
            meta.lookup("q_limbs range check", [(qa(l, cur), c.table)])
        for l in c.r_limbs:
            meta.lookup("r_limbs range check", [(qa(l, cur), c.table)])
        return c


def poseidon_parameters():
    """round constants (RC[round * 5 + i]) and MDS of the reference's BN254 Poseidon (bn245_poseidon/constants.rs), Grain-generated"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from poseidon_grain import bn254_t5
    return bn254_t5()


class PoseidonBn254ChipConfig:
    @staticmethod
    def configure(meta):
        c = PoseidonBn254ChipConfig()
        c.state = [meta.advice_column() for _ in range(T)]
        c.constants = [meta.fixed_column() for _ in range(T)]
        c.q_f, c.q_p = meta.selector(), meta.selector()
        for s in c.state:
            meta.enable_equality(s)
        c.rc, c.mds = poseidon_parameters()
        cur, nxt = h2.Rotation.cur(), h2.Rotation.next()

        def round_gate(selector, full):
            nxt_state = [meta.query_advice(s, nxt) for s in c.state]
            state = [meta.query_advice(s, cur) for s in c.state]
            consts = [meta.query_fixed(k, cur) for k in c.constants]
            q = meta.query_selector(selector)
            after = [s + k for s, k in zip(state, consts)]
            for i in range(T if full else 1):
                x = after[i]
                after[i] = x * x * x * x * x
            out = []
            for i in range(T):
                acc = h2.Expression.constant(0)
                for j in range(T):
                    acc = acc + after[j] * c.mds[i][j]
                out.append(q * (nxt_state[i] - acc))
            return out
        meta.create_gate("partial round", round_gate(c.q_p, False))
        meta.create_gate("full round", round_gate(c.q_f, True))
        return c


class AllChipConfig:
    @staticmethod
    def configure(meta, table_bits=16, modulus=GOLDILOCKS_MODULUS):
        c = AllChipConfig()
        c.arithmetic_config = ArithmeticChipConfig.configure(meta, table_bits, modulus)
        c.poseidon_config = PoseidonBn254ChipConfig.configure(meta)
        return c


# ---- assignment -----------------------------------------------------------------------------------------------------------------------
def mul_add_divmod_p(a, b, c):
    """(q, r) with a b + c = q p + r, 0 <= r < p, for uint64 arrays a, b, c < 2^64 (ArithmeticChip::assign's div_rem, vectorised):
    128-bit product from 32-bit halves, then three folds of 2^64 = p + (2^32 - 1)"""
    a, b, c = (np.asarray(v, dtype=np.uint64) for v in (a, b, c))
    m32 = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    al, ah, bl, bh = a & m32, a >> s32, b & m32, b >> s32
    ll, lh, hl, hh = al * bl, al * bh, ah * bl, ah * bh
    mid = (ll >> s32) + (lh & m32) + (hl & m32)
    lo = (ll & m32) | ((mid & m32) << s32)
    hi = hh + (lh >> s32) + (hl >> s32) + (mid >> s32)
    lo2 = lo + c
    hi = hi + (lo2 < lo).astype(np.uint64)
    lo = lo2
    P = np.uint64(GOLDILOCKS_MODULUS)
    eps = np.uint64(0xFFFFFFFF)
    qlo, qhi = hi.copy(), np.zeros_like(hi)          # quotient accumulates: T = hi 2^64 + lo = hi p + (hi eps + lo)

    # hi * eps exactly: hi * (2^32 - 1) = hi_h 2^64 + ... ; do it with the same 32-bit splitting
    for _ in range(3):
        hl_, hh_ = hi & m32, hi >> s32
        p_lo = hl_ * eps                              # < 2^64
        p_hi = hh_ * eps                              # weight 2^32
        t_lo = p_lo + ((p_hi & m32) << s32)
        carry = (t_lo < p_lo).astype(np.uint64)
        t_hi = (p_hi >> s32) + carry
        n_lo = t_lo + lo
        t_hi = t_hi + (n_lo < t_lo).astype(np.uint64)
        hi, lo = t_hi, n_lo
        n_q = qlo + hi
        qhi = qhi + (n_q < qlo).astype(np.uint64)
        qlo = n_q
    assert not hi.any()
    over = lo >= P
    lo = np.where(over, lo - P, lo)
    n_q = qlo + over.astype(np.uint64)
    qhi = qhi + (n_q < qlo).astype(np.uint64)
    return (qhi, n_q), lo


def mul_add_divmod(a, b, c, modulus):
    """-> ((q_hi, q_lo), r) for operands below the modulus"""
    if modulus == GOLDILOCKS_MODULUS:
        return mul_add_divmod_p(a, b, c)
    assert modulus < (1 << 31)
    t = np.asarray(a, dtype=np.uint64) * np.asarray(b, dtype=np.uint64) + np.asarray(c, dtype=np.uint64)
    q = t // np.uint64(modulus)
    return (np.zeros_like(q), q), t - q * np.uint64(modulus)


def limbs_of(v_lo, v_hi, count, bits):
    """`count` little-endian limbs of `bits` bits of the up-to-128-bit values (v_hi 2^64 + v_lo)"""
    out = []
    mask = np.uint64((1 << bits) - 1)
    for i in range(count):
        sh = i * bits
        if sh < 64:
            limb = (v_lo >> np.uint64(sh)) & mask
            if sh + bits > 64:
                limb = limb | ((v_hi << np.uint64(64 - sh)) & mask)
        else:
            limb = (v_hi >> np.uint64(sh - 64)) & mask
        out.append(limb)
    return out


class Witness:
    """advice / fixed columns of a 2^k-row circuit as [columns][n][4] uint64 arrays (little-endian limbs) + the copy constraints"""

    def __init__(self, cs, k):
        self.cs, self.k, self.n = cs, k, 1 << k
        self.advice = np.zeros((cs.num_advice, self.n, 4), dtype=np.uint64)
        self.fixed = np.zeros((cs.num_fixed, self.n, 4), dtype=np.uint64)
        self.instance = [[] for _ in range(cs.num_instance)]
        self.assembly = h2.Assembly(self.n, cs.permutation)
        self.usable = self.n - (cs.blinding_factors() + 1)

    def set_int(self, arr, col, row, v):
        v = int(v) % R
        for j in range(4):
            arr[col, row, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF

    def advice_ints(self):
        return [h2.from_limbs(self.advice[c]) for c in range(self.cs.num_advice)]

    def fixed_ints(self):
        return [h2.from_limbs(self.fixed[c]) for c in range(self.cs.num_fixed)]


def assign_arithmetic_rows(w, cfg, row0, a, b, c):
    """ArithmeticChip::assign on rows row0 .. row0 + len(a): a b + c = q p + r with the limb decompositions, s_base and s_limb on"""
    a, b, c = (np.asarray(v, dtype=np.uint64) for v in (a, b, c))
    m = a.shape[0]
    rows = slice(row0, row0 + m)
    (qhi, qlo), r = mul_add_divmod(a, b, c, cfg.modulus)
    A = w.advice
    A[cfg.a.index, rows, 0], A[cfg.b.index, rows, 0], A[cfg.c.index, rows, 0] = a, b, c
    A[cfg.q.index, rows, 0], A[cfg.q.index, rows, 1] = qlo, qhi
    A[cfg.r.index, rows, 0] = r
    for col, limb in zip(cfg.q_limbs, limbs_of(qlo, qhi, Q_LIMBS, cfg.table_bits)):
        A[col.index, rows, 0] = limb
    for col, limb in zip(cfg.r_limbs, limbs_of(r, np.zeros_like(r), 4, cfg.table_bits)):
        A[col.index, rows, 0] = limb
    w.fixed[cfg.s_base.index, rows, 0] = 1
    w.fixed[cfg.s_limb.index, rows, 0] = 1
    return r


def assign_value_rows(w, cfg, row0, values):
    """ArithmeticChip::assign_value: r = the value, q = p - r, s_limb and s_range on (0 <= r < p because q's limbs exist)"""
    r = np.asarray(values, dtype=np.uint64)
    rows = slice(row0, row0 + r.shape[0])
    q = np.uint64(cfg.modulus) - r
    A = w.advice
    A[cfg.q.index, rows, 0], A[cfg.r.index, rows, 0] = q, r
    for col, limb in zip(cfg.q_limbs, limbs_of(q, np.zeros_like(q), Q_LIMBS, cfg.table_bits)):
        A[col.index, rows, 0] = limb
    for col, limb in zip(cfg.r_limbs, limbs_of(r, np.zeros_like(r), 4, cfg.table_bits)):
        A[col.index, rows, 0] = limb
    w.fixed[cfg.s_limb.index, rows, 0] = 1
    w.fixed[cfg.s_range.index, rows, 0] = 1


def assign_permutation(w, cfg, row0, state):
    """PoseidonBn254Chip::apply_permute: the state on row0, one row per round (round constants in the fixed columns, q_f / q_p on),
    the final state on row0 + 68.  Returns the final state."""
    s = [int(x) % R for x in state]
    rc, mds = cfg.rc, cfg.mds
    for rnd in range(R_F + R_P):
        row = row0 + rnd
        full = rnd < R_F // 2 or rnd >= R_F // 2 + R_P
        for i in range(T):
            w.set_int(w.advice, cfg.state[i].index, row, s[i])
            w.set_int(w.fixed, cfg.constants[i].index, row, rc[rnd * T + i])
        w.fixed[(cfg.q_f if full else cfg.q_p).index, row, 0] = 1
        t = [(s[i] + rc[rnd * T + i]) % R for i in range(T)]
        if full:
            t = [pow(x, 5, R) for x in t]
        else:
            t[0] = pow(t[0], 5, R)
        s = [sum(mds[i][j] * t[j] for j in range(T)) % R for i in range(T)]
    for i in range(T):
        w.set_int(w.advice, cfg.state[i].index, row0 + R_F + R_P, s[i])
    return s


def synthetic_circuit(k, table_bits=16, seed=0x355, n_permutations=None, n_arith=None, modulus=None):
    """A 2^k-row instance over AllChipConfig: the range table in the first 2^table_bits rows of the table column; `n_arith` rows of
    ArithmeticChip::assign with random operands (default: every usable row but a band of assign_value rows), the r of one row copied into
    the a of the next for a stretch (equality constraints), two of the results exposed through the instance column; `n_permutations`
    chained Poseidon permutations (69 rows each) down the state columns, random values in the state columns below them (selectors off).
    -> (cs, config, Witness)"""
    if modulus is None:                                     # the reference's p with its 16-bit limbs; a prime just under 2^(4 t) otherwise
        modulus = GOLDILOCKS_MODULUS if table_bits >= 16 else ((1 << (4 * table_bits)) - {4: 15, 5: 3, 6: 3, 7: 57}[table_bits] if table_bits <= 7 else (1 << 31) - 1)
    cs = h2.ConstraintSystem()
    cfg = AllChipConfig.configure(cs, table_bits, modulus)
    ar, po = cfg.arithmetic_config, cfg.poseidon_config
    w = Witness(cs, k)
    n, u = w.n, w.usable
    assert (1 << table_bits) <= u, "the %d-bit range table does not fit 2^%d rows" % (table_bits, k)
    rng = np.random.default_rng(seed)
    w.fixed[ar.table.index, :1 << table_bits, 0] = np.arange(1 << table_bits, dtype=np.uint64)
    # arithmetic rows
    n_val = min(64, u // 8)
    n_ar = u - n_val if n_arith is None else min(n_arith, u - n_val)
    P = modulus
    a = rng.integers(0, P, n_ar, dtype=np.uint64)
    b = rng.integers(0, P, n_ar, dtype=np.uint64)
    c = rng.integers(0, P, n_ar, dtype=np.uint64)
    chain = min(n_ar - 1, 2048)
    # rows 1 .. chain take the previous row's r as their a (a chain of copy constraints): sequential by nature, so a short stretch
    for i in range(chain):
        (_, _), r_i = mul_add_divmod(a[i:i + 1], b[i:i + 1], c[i:i + 1], P)
        a[i + 1] = r_i[0]
    r = assign_arithmetic_rows(w, ar, 0, a, b, c)
    for i in range(chain):
        w.assembly.copy(ar.r, i, ar.a, i + 1)
    assign_value_rows(w, ar, n_ar, rng.integers(0, P, n_val, dtype=np.uint64))
    # a constant through the `constant` column (ArithmeticChip::assign_constant): row n_ar - 1's c := the constant 7 ... keep it simple: the
    # constant column holds c of row 0 and is tied to it
    w.fixed[ar.constant.index, 0] = w.advice[ar.c.index, 0]
    w.assembly.copy(ar.c, 0, ar.constant, 0)
    # public inputs: r of rows 0 and 1
    w.instance[ar.instance.index] = [int(r[0]), int(r[1])]
    w.assembly.copy(ar.r, 0, ar.instance, 0)
    w.assembly.copy(ar.r, 1, ar.instance, 1)
    # Poseidon permutations, chained through copy constraints (apply_permute re-assigns the state and constrains it equal)
    rows_per = R_F + R_P + 1
    n_perm = (min(u // rows_per, 8) if n_permutations is None else min(n_permutations, u // rows_per))
    state = [int(x) for x in rng.integers(0, 1 << 62, T)]
    for p_i in range(n_perm):
        row0 = p_i * rows_per
        out = assign_permutation(w, po, row0, state)
        if p_i:
            for i in range(T):
                w.assembly.copy(po.state[i], row0 - 1, po.state[i], row0)
        state = out
    lo = n_perm * rows_per
    if lo < u:                                              # unconstrained cells below: random field elements (what a dense witness looks like to the MSMs)
        junk = rng.integers(0, 1 << 63, (T, u - lo, 4), dtype=np.uint64)
        junk[:, :, 3] >>= np.uint64(2)
        for i in range(T):
            w.advice[po.state[i].index, lo:u] = junk[i]
    return cs, cfg, w
