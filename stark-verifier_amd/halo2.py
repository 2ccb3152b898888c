"""Host-side mirror of the halo2 front end the reference's SNARK finalisation goes through:
`ConstraintSystem` / `Expression` / `Rotation` / the permutation `Assembly` as `Circuit::configure` and `keygen_vk` / `keygen_pk` use
them (src/plonky2_verifier/verifier_api.rs:77-92; the reference's own gates and lookups: chip/native_chip/arithmetic_chip.rs:44-160,
poseidon_bn254_chip.rs:27-123), reduced to what the prover's data-parallel stages need: the column / query / gate / lookup /
permutation description of a circuit, compiled to the descriptor blob `gl355_plonk_keygen` takes (include/gl355.h) -- gates and
lookup expressions as small register programs.

halo2_proofs itself is an un-vendored dependency of the reference (Cargo.lock); names and semantics follow its published API
(`meta.advice_column()`, `meta.create_gate`, `meta.lookup`, `meta.enable_equality`, `Rotation::cur/next/prev`, `cs.degree()`,
`cs.blinding_factors()`).  Deviations, all on the host side of the boundary: selectors are plain fixed columns (halo2 compresses
simple selectors into fewer fixed columns at keygen), one phase, no challenges-as-expressions.
"""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617     # bn256::Fr
S = 28
GENERATOR = 7
ROOT_OF_UNITY = pow(GENERATOR, (R - 1) >> S, R)
DELTA = pow(GENERATOR, 1 << S, R)                      # Fr::DELTA: generator of the t-order subgroup, separates the permutation's column cosets
ZETA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23        # Fr::ZETA, a primitive cube root of unity: the extended domain's coset shift
assert pow(ZETA, 3, R) == 1 and ZETA != 1

ADVICE, FIXED, INSTANCE = 0, 1, 2
KIND_NAMES = {ADVICE: "advice", FIXED: "fixed", INSTANCE: "instance"}


class Rotation(int):
    @staticmethod
    def cur():
        return Rotation(0)

    @staticmethod
    def next():
        return Rotation(1)

    @staticmethod
    def prev():
        return Rotation(-1)


class Column:
    def __init__(self, kind, index):
        self.kind, self.index = kind, index

    def __eq__(self, o):
        return isinstance(o, Column) and (self.kind, self.index) == (o.kind, o.index)

    def __hash__(self):
        return hash((self.kind, self.index))

    def __repr__(self):
        return "%s[%d]" % (KIND_NAMES[self.kind], self.index)


class Expression:
    """plonk::Expression: Constant | Query (advice / fixed / instance at a rotation) | Negated | Sum | Product | Scaled"""
    __slots__ = ("op", "a", "b")

    def __init__(self, op, a=None, b=None):
        self.op, self.a, self.b = op, a, b

    @staticmethod
    def constant(v):
        return Expression("const", int(v) % R)

    @staticmethod
    def _wrap(x):
        return x if isinstance(x, Expression) else Expression.constant(x)

    def __add__(self, o):
        return Expression("add", self, Expression._wrap(o))

    __radd__ = __add__

    def __sub__(self, o):
        return Expression("add", self, Expression("neg", Expression._wrap(o)))

    def __rsub__(self, o):
        return Expression._wrap(o) - self

    def __neg__(self):
        return Expression("neg", self)

    def __mul__(self, o):
        if isinstance(o, Expression):
            return Expression("mul", self, o)
        return Expression("scale", self, int(o) % R)

    __rmul__ = __mul__

    def degree(self):
        if self.op == "const":
            return 0
        if self.op == "query":
            return 1
        if self.op in ("neg", "scale"):
            return self.a.degree()
        if self.op == "add":
            return max(self.a.degree(), self.b.degree())
        return self.a.degree() + self.b.degree()

    def evaluate(self, query):
        """tree evaluation over Python integers; query(kind, query_index) -> value"""
        op = self.op
        if op == "const":
            return self.a
        if op == "query":
            return query(self.a, self.b)
        if op == "neg":
            return (-self.a.evaluate(query)) % R
        if op == "scale":
            return self.a.evaluate(query) * self.b % R
        if op == "add":
            return (self.a.evaluate(query) + self.b.evaluate(query)) % R
        return self.a.evaluate(query) * self.b.evaluate(query) % R


class ConstraintSystem:
    def __init__(self):
        self.num_advice = self.num_fixed = self.num_instance = 0
        self.selectors = []                      # fixed-column indices that were declared as selectors (bookkeeping only)
        self.queries = {ADVICE: [], FIXED: [], INSTANCE: []}     # per kind: [(column index, rotation)] in first-use order
        self.gates = []                          # (name, [Expression])
        self.lookups = []                        # (name, [input Expression], [table Expression])
        self.permutation = []                    # columns with equality enabled, in enable order
        self.minimum_degree = None

    # ---- columns -----------------------------------------------------------------------------------------------------
    def advice_column(self):
        self.num_advice += 1
        return Column(ADVICE, self.num_advice - 1)

    def fixed_column(self):
        self.num_fixed += 1
        return Column(FIXED, self.num_fixed - 1)

    def instance_column(self):
        self.num_instance += 1
        return Column(INSTANCE, self.num_instance - 1)

    def selector(self):
        c = self.fixed_column()
        self.selectors.append(c.index)
        return c

    def lookup_table_column(self):
        return self.fixed_column()

    def enable_equality(self, column):
        self._query(column, 0)                   # halo2: query_any_index(column, Rotation::cur()) -- the argument reads the column at x
        if column not in self.permutation:
            self.permutation.append(column)

    def enable_constant(self, column):
        self.enable_equality(column)

    # ---- queries -----------------------------------------------------------------------------------------------------
    def _query(self, column, rotation):
        q = self.queries[column.kind]
        key = (column.index, int(rotation))
        if key not in q:
            q.append(key)
        return Expression("query", column.kind, q.index(key))

    def query_advice(self, column, rotation=0):
        assert column.kind == ADVICE
        return self._query(column, rotation)

    def query_fixed(self, column, rotation=0):
        assert column.kind == FIXED
        return self._query(column, rotation)

    query_selector = query_fixed

    def query_instance(self, column, rotation=0):
        assert column.kind == INSTANCE
        return self._query(column, rotation)

    def query_any(self, column, rotation=0):
        return self._query(column, rotation)

    # ---- gates / lookups ---------------------------------------------------------------------------------------------
    def create_gate(self, name, polys):
        polys = list(polys)
        assert polys
        self.gates.append((name, polys))

    def lookup(self, name, pairs):
        pairs = list(pairs)
        ins = [p[0] for p in pairs]
        tabs = [p[1] if isinstance(p[1], Expression) else self.query_fixed(p[1]) for p in pairs]
        self.lookups.append((name, ins, tabs))

    # ---- derived shape (plonk::ConstraintSystem::degree / blinding_factors, permutation::Argument, lookup::Argument) --
    def degree(self):
        d = 3                                                           # permutation::Argument::required_degree: 3 with or without columns
        for _, ins, tabs in self.lookups:                               # lookup::Argument::required_degree
            di = max([1] + [e.degree() for e in ins])
            dt = max([1] + [e.degree() for e in tabs])
            d = max(d, 4, 2 + di + dt)
        for _, polys in self.gates:
            d = max([d] + [p.degree() for p in polys])
        return max(d, self.minimum_degree or 1)

    def blinding_factors(self):
        per_col = {}
        for col, _ in self.queries[ADVICE]:
            per_col[col] = per_col.get(col, 0) + 1
        factors = max([1] + list(per_col.values()))
        return max(3, factors) + 2

    def minimum_rows(self):
        return self.blinding_factors() + 3

    def chunk_len(self):
        return self.degree() - 2                                        # columns per permutation product polynomial

    def all_gate_polys(self):
        return [p for _, polys in self.gates for p in polys]


class Assembly:
    """permutation::keygen::Assembly: the copy-constraint cycles over the equality-enabled columns (halo2 book, "Permutation argument"):
    mapping[column][row] = the next cell of the cycle, aux = a representative of the cycle, sizes = its length (numpy arrays: a k = 23
    circuit has 10^8 cells)"""

    def __init__(self, n, columns):
        self.n, self.columns = n, list(columns)
        m = len(self.columns)
        ident = np.empty((m, n, 2), dtype=np.uint32)
        ident[:, :, 0] = np.arange(m, dtype=np.uint32)[:, None]
        ident[:, :, 1] = np.arange(n, dtype=np.uint32)[None, :]
        self.mapping = ident
        self.aux = ident.copy()
        self.sizes = np.ones((m, n), dtype=np.uint32)

    def copy(self, left_column, left_row, right_column, right_row):
        lc, rc = self.columns.index(left_column), self.columns.index(right_column)
        left, right = tuple(int(v) for v in self.aux[lc, left_row]), tuple(int(v) for v in self.aux[rc, right_row])
        if left == right:
            return
        if self.sizes[left] < self.sizes[right]:
            left, right = right, left
        self.sizes[left] += self.sizes[right]
        i = right
        while True:
            self.aux[i] = left
            i = tuple(int(v) for v in self.mapping[i])
            if i == right:
                break
        tmp = self.mapping[lc, left_row].copy()
        self.mapping[lc, left_row] = self.mapping[rc, right_row]
        self.mapping[rc, right_row] = tmp

    def mapping_array(self):
        """[n_perm][n][2] uint32: (column position in the permutation, row) each cell maps to"""
        return self.mapping


# ---- compilation of expressions to the register programs of the descriptor ------------------------------------------------------
OP_ADD, OP_SUB, OP_MUL, OP_EMIT, OP_NEG, OP_MOV = range(6)
K_REG, K_CONST, K_ADVICE, K_FIXED, K_INSTANCE = range(5)
MAX_REGS = 12
_KIND_OPERAND = {ADVICE: K_ADVICE, FIXED: K_FIXED, INSTANCE: K_INSTANCE}


class _Node:
    __slots__ = ("op", "kids", "val", "uses", "reg", "need")

    def __init__(self, op, kids, val):
        self.op, self.kids, self.val, self.uses, self.reg, self.need = op, kids, val, 0, None, 0


class ProgramBuilder:
    """Expression trees -> straight-line code over at most MAX_REGS 256-bit registers.  An instruction is four u32:
    (op, dst register, operand a, operand b), an operand = kind << 24 | index with kind in REG / CONST (pool index) / ADVICE / FIXED /
    INSTANCE (query index).  EMIT a: "this value is the next polynomial" -- the evaluator folds it into its running sum
    (sum * y + a for constraints, sum * theta + a for the expressions of a lookup).  Structurally equal subexpressions are computed
    once (halo2's GraphEvaluator does the same) and stay in their register until their last use."""

    def __init__(self, const_pool):
        self.code, self.pool, self.free, self.peak = [], const_pool, list(range(MAX_REGS - 1, -1, -1)), 0
        self.nodes, self.roots, self._memo = {}, [], {}

    def _const(self, v):
        v %= R
        if v not in self.pool:
            self.pool.append(v)
        return (K_CONST << 24) | self.pool.index(v)

    def _dag(self, e):
        m = self._memo.get(id(e))
        if m is not None:
            return m
        if e.op == "const":
            key, kids, val = ("c", e.a), (), e.a
        elif e.op == "query":
            key, kids, val = ("q", e.a, e.b), (), (e.a, e.b)
        elif e.op == "neg":
            k = self._dag(e.a)
            key, kids, val = ("n", id(k)), (k,), None
        elif e.op == "scale":
            k = self._dag(e.a)
            key, kids, val = ("s", id(k), e.b), (k,), e.b
        elif e.op == "add" and e.b.op == "neg":
            x, y = self._dag(e.a), self._dag(e.b.a)
            key, kids, val = ("-", id(x), id(y)), (x, y), None
        else:
            x, y = self._dag(e.a), self._dag(e.b)
            key, kids, val = ("+" if e.op == "add" else "*", id(x), id(y)), (x, y), None
        node = self.nodes.get(key)
        if node is None:
            node = self.nodes[key] = _Node(key[0], kids, val)
            node.need = 0 if not kids else max(1, max(k.need for k in kids) if len(kids) == 1 or kids[0].need != kids[1].need else kids[0].need + 1)
            for k in kids:
                k.uses += 1
        self._memo[id(e)] = node
        return node

    def emit(self, e):
        node = self._dag(e)
        node.uses += 1
        self.roots.append(node)

    def _alloc(self):
        if not self.free:
            raise ValueError("expression needs more than %d registers" % MAX_REGS)
        r = self.free.pop()
        self.peak = max(self.peak, MAX_REGS - len(self.free))
        return r

    def _done_with(self, node):
        node.uses -= 1
        if node.uses == 0 and node.reg is not None:
            self.free.append(node.reg)
            node.reg = None

    def _gen(self, node):
        if node.op == "c":
            return self._const(node.val)
        if node.op == "q":
            return (_KIND_OPERAND[node.val[0]] << 24) | node.val[1]
        if node.reg is not None:
            return (K_REG << 24) | node.reg
        kids = node.kids
        if len(kids) == 2 and kids[1].need > kids[0].need and kids[0] is not kids[1]:
            b = self._gen(kids[1])
            a = self._gen(kids[0])
        else:
            a = self._gen(kids[0])
            b = self._gen(kids[1]) if len(kids) == 2 else 0
        for k in kids:
            self._done_with(k)
        d = self._alloc()
        if node.op == "n":
            self.code.append((OP_NEG, d, a, 0))
        elif node.op == "s":
            self.code.append((OP_MUL, d, a, self._const(node.val)))
        else:
            self.code.append(({"+": OP_ADD, "-": OP_SUB, "*": OP_MUL}[node.op], d, a, b))
        node.reg = d
        return (K_REG << 24) | d

    def finish(self):
        if self.roots is None:
            return
        for node in self.roots:
            v = self._gen(node)
            self.code.append((OP_EMIT, 0, v, 0))
            self._done_with(node)
        self.roots = None

    def words(self):
        self.finish()
        return np.array(self.code, dtype=np.uint32).reshape(-1, 4)


MAGIC = 0x4B4C503535334C47          # "GL355PLK" little endian


def to_limbs(values):
    """Python integers -> [len][4] uint64 (little-endian limbs)"""
    out = np.zeros((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        v = int(v) % R
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_limbs(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def vk_digest(cs, k, fixed_commitments, sigma_commitments, keccak256):
    """the scalar the transcript starts from (halo2: vk.transcript_repr, a hash of the PINNED verifying key): Keccak-256 over the descriptor
    (digest field zero: shape, queries, permutation columns, gate and lookup programs) and the key's fixed and permutation commitments
    ([cols][8] uint64: x, y little-endian words; the identity = zeros), read as a big-endian integer mod r -- what gl355_plonk_keygen derives
    itself (include/gl355.h).  A stand-in with transcript_repr's role (halo2's Blake2b-of-debug-string cannot be restated); it binds the
    fixed values and the copy constraints through their commitments.  keccak256: bytes -> 32 bytes (the caller's implementation: the
    checker recomputes the digest with its own)."""
    pre = export_desc(cs, k, 0).tobytes()
    pre += np.ascontiguousarray(fixed_commitments, dtype=np.uint64).tobytes() + np.ascontiguousarray(sigma_commitments, dtype=np.uint64).tobytes()
    return int.from_bytes(keccak256(pre), "big") % R


def export_desc(cs, k, digest):
    """the descriptor blob of include/gl355.h (gl355_plonk_keygen): header, permutation columns, queries, constant pool, gate program,
    per lookup the input and the table program"""
    pool = []
    gate = ProgramBuilder(pool)
    for p in cs.all_gate_polys():
        gate.emit(p)
    lks = []
    for _, ins, tabs in cs.lookups:
        pi, pt = ProgramBuilder(pool), ProgramBuilder(pool)
        for e in ins:
            pi.emit(e)
        for e in tabs:
            pt.emit(e)
        lks.append((pi, pt))
    hdr = np.zeros(24, dtype=np.uint64)
    hdr[0], hdr[1], hdr[2] = MAGIC, 1, k
    hdr[3], hdr[4], hdr[5] = cs.num_advice, cs.num_fixed, cs.num_instance
    hdr[6], hdr[7], hdr[8], hdr[9] = len(cs.permutation), len(cs.lookups), cs.degree(), cs.blinding_factors()
    hdr[10], hdr[11], hdr[12] = len(cs.queries[ADVICE]), len(cs.queries[FIXED]), len(cs.queries[INSTANCE])
    for pb in [gate] + [x for pair in lks for x in pair]:
        pb.finish()
    hdr[13], hdr[14], hdr[15] = len(pool), len(gate.code), len(cs.all_gate_polys())
    hdr[16:20] = to_limbs([digest])[0]
    parts = [hdr]
    parts.append(np.array([(c.kind << 32) | c.index for c in cs.permutation], dtype=np.uint64))
    for kind in (ADVICE, FIXED, INSTANCE):
        parts.append(np.array([(col << 32) | (rot & 0xFFFFFFFF) for col, rot in cs.queries[kind]], dtype=np.uint64))
    parts.append(to_limbs(pool).reshape(-1))

    def prog_words(pb):
        w = pb.words().reshape(-1)
        return w.view(np.uint64) if w.size else np.zeros(0, dtype=np.uint64)
    parts.append(prog_words(gate))
    for pi, pt in lks:
        parts.append(np.array([len(pi.code), len(pt.code)], dtype=np.uint64))
        parts.append(prog_words(pi))
        parts.append(prog_words(pt))
    return np.concatenate([np.ascontiguousarray(p, dtype=np.uint64).reshape(-1) for p in parts])


# ---- the prover behind the C ABI (include/gl355.h: gl355_plonk_keygen / gl355_plonk_prove) -----------------------------------------------
STAGES = ("advice", "lookup_permute", "permutation", "lookup_product", "vanishing_random", "evaluate_h", "quotient_commit", "evaluations", "shplonk")


def kzg_setup(ctx, k, tau, lagrange=True):
    """ParamsKZG::<Bn256>::setup(k, rng) (verifier_api.rs:77) with the secret handed in: (g, g_lagrange) as [2^k][8] uint64 arrays"""
    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64) if lagrange else None
    t = to_limbs([tau])[0]
    ctx.check(ctx.lib.gl355_kzg_setup(ctx.h, t.ctypes.data, k, g.ctypes.data, gl.ctypes.data if lagrange else None))
    return g, gl


class PlonkProver:
    """keygen_pk + create_proof of one circuit on one GPU context (chip/native_chip/test_utils.rs:57-95 through gl355_plonk_*).
    g / g_lagrange: numpy arrays (copied to the device) or device pointers (ints; must outlive the prover)."""

    def __init__(self, ctx, cs, k, g, g_lagrange, fixed, mapping, digest=None):
        import ctypes as C
        self.ctx, self.cs, self.k, self.n = ctx, cs, k, 1 << k
        self.desc = export_desc(cs, k, 0 if digest is None else digest)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint64)
        mapping = np.ascontiguousarray(mapping, dtype=np.uint32)
        assert fixed.shape == (cs.num_fixed, self.n, 4) and mapping.shape == (len(cs.permutation), self.n, 2)
        self._keep = (g, g_lagrange)
        ptr = lambda a: a if isinstance(a, int) else a.ctypes.data       # noqa: E731
        self.h = C.c_void_p()
        ctx.check(ctx.lib.gl355_plonk_keygen(ctx.h, self.desc.ctypes.data, self.desc.size, ptr(g), ptr(g_lagrange), fixed.ctypes.data, mapping.ctypes.data, C.byref(self.h)))
        info = np.zeros(8, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_plonk_pk_info(self.h, info.ctypes.data))
        self.info = dict(k=int(info[0]), extended_k=int(info[1]), n_sets=int(info[2]), n_pieces=int(info[3]), usable=int(info[4]), proof_bytes=int(info[5]))
        self.fixed_commitments = np.zeros((cs.num_fixed, 8), dtype=np.uint64)
        self.sigma_commitments = np.zeros((len(cs.permutation), 8), dtype=np.uint64)
        ctx.check(ctx.lib.gl355_plonk_pk_commitments(self.h, self.fixed_commitments.ctypes.data, self.sigma_commitments.ctypes.data))
        if digest is None:          # derived by keygen from the descriptor and the commitments above
            d = np.zeros(4, dtype=np.uint64)
            ctx.check(ctx.lib.gl355_plonk_pk_digest(self.h, d.ctypes.data))
            self.digest = from_limbs(d)[0]
        else:
            self.set_digest(digest)

    def set_digest(self, digest):
        d = to_limbs([digest])[0]
        self.ctx.check(self.ctx.lib.gl355_plonk_pk_set_digest(self.h, d.ctypes.data))
        self.digest = digest

    def prove(self, advice, instances, seed, want_trace=False, timed=False):
        """advice: [num_advice][n][4] uint64 (numpy, or a device pointer as int); instances: per instance column a list of integers;
        seed: 32 bytes.  -> proof bytes (, trace dict)(, {stage: ms})"""
        import ctypes as C
        ctx = self.ctx
        if not isinstance(advice, int):
            advice = np.ascontiguousarray(advice, dtype=np.uint64)
            assert advice.shape == (self.cs.num_advice, self.n, 4)
        flat = to_limbs([v for col in instances for v in col]) if any(len(c) for c in instances) else np.zeros((1, 4), dtype=np.uint64)
        lens = np.array([len(c) for c in instances] + [0], dtype=np.uint32)
        buf = np.zeros(self.info["proof_bytes"], dtype=np.uint8)
        out_len = C.c_uint64(0)
        trace = np.zeros(32, dtype=np.uint64)
        ms = np.zeros(len(STAGES), dtype=np.float64)
        seed = bytes(seed)
        assert len(seed) == 32
        ctx.check(ctx.lib.gl355_plonk_prove(ctx.h, self.h, advice if isinstance(advice, int) else advice.ctypes.data, flat.ctypes.data, lens.ctypes.data, seed,
                                            buf.ctypes.data, buf.size, C.byref(out_len), trace.ctypes.data if want_trace else None, ms.ctypes.data if timed else None))
        out = [bytes(buf[:out_len.value])]
        if want_trace:
            t = from_limbs(trace.reshape(8, 4))
            out.append(dict(zip(("theta", "beta", "gamma", "y", "x", "shplonk_y", "shplonk_v", "shplonk_u"), t)))
        if timed:
            out.append(dict(zip(STAGES, (float(v) for v in ms))))
        return out[0] if len(out) == 1 else tuple(out)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.gl355_plonk_pk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
