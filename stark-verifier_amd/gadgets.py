"""Target-level circuit builder with eager witness generation, and the gadget set the recursive
verifier needs (host side; mirrors the plonky2 `CircuitBuilder` API the reference uses at
src/plonky2_semaphore/recursion.rs:49-168 and wrapper.rs:35-47: add_virtual_*, connect, constant,
register_public_inputs, hash_n_to_hash_no_pad, verify_proof, build).

Design: every target is a wire (row, col) of some gate row and carries its VALUE -- building the
circuit for a concrete input and generating its witness are one pass ("eager witness").  The layout
(gate rows, op slots, copy constraints) depends only on the sequence of builder calls, never on the
values, so the first pass fixes selectors / sigmas (CircuitData) and every later pass with other
inputs reproduces the same layout and only yields a new witness (checked by a structure hash).

Gates emitted (all evaluated by csrc/quotient.hip and restated in tests/plonk_verifier.py; wire
layouts from the reference's chip/plonk/gates/*.rs): Constant{2}, PublicInput, Noop (free inputs),
Arithmetic{20}, ArithmeticExtension{10}, Poseidon, PoseidonMds, BaseSum{32}, RandomAccess{4,4,2},
Reducing{43}, ReducingExtension{32}.
"""
import hashlib

import numpy as np

from ._lib import (GATE_ARITHMETIC, GATE_ARITHMETIC_EXT, GATE_BASE_SUM, GATE_CONSTANT, GATE_NOOP, GATE_POSEIDON,
                   GATE_POSEIDON_MDS, GATE_PUBLIC_INPUT, GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT)
from .plonk import CircuitBuilder, CircuitConfig, P, host_hash_no_pad, poseidon_gate_witness

RA_PARAM = 4 | 4 << 8 | 2 << 16
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]

# witness-tape opcodes (include/gl355.h GL355_TAPE_*, interpreter: csrc/witness_tape.cpp).  Every builder
# primitive appends the entries that recompute its wires from earlier wires, so a circuit built once can be
# re-witnessed for new inputs by gl355_witness_replay without running the Python gadgets again.
(TAPE_CONST, TAPE_INPUT, TAPE_COPY, TAPE_ASSERT_EQ, TAPE_ARITH, TAPE_ARITH_EXT, TAPE_POSEIDON, TAPE_MDS_EXT,
 TAPE_BASE_SUM, TAPE_RANDOM_ACCESS, TAPE_REDUCING, TAPE_LO32, TAPE_HI32, TAPE_EXT_INV) = range(14)
# which of the four operand fields of an entry are wire references (bit i = operand i)
TAPE_WIRE_FIELDS = {TAPE_CONST: 1, TAPE_INPUT: 1, TAPE_COPY: 3, TAPE_ASSERT_EQ: 3, TAPE_ARITH: 1, TAPE_ARITH_EXT: 1,
                    TAPE_POSEIDON: 1, TAPE_MDS_EXT: 1, TAPE_BASE_SUM: 1, TAPE_RANDOM_ACCESS: 1, TAPE_REDUCING: 1,
                    TAPE_LO32: 3, TAPE_HI32: 3, TAPE_EXT_INV: 15}


class T:
    """a base-field target: home wire + value"""
    __slots__ = ("row", "col", "v")

    def __init__(self, row, col, v):
        self.row, self.col, self.v = row, col, v % P


def emul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def einv(a):
    norm = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    ni = pow(norm, P - 2, P)
    return (a[0] * ni % P, (-a[1]) * ni % P)


class GadgetBuilder:
    def __init__(self, config=None):
        self.cb = CircuitBuilder(config or CircuitConfig())
        self.nw = self.cb.config.num_wires
        self.routed = self.cb.config.num_routed_wires
        self.rows = {}            # row -> {col: value}   (sparse witness)
        self.open = {}            # gate key -> (row, next slot)
        self.consts = {}          # value -> T
        self.public_inputs = []
        self.trace = hashlib.sha256()
        self._free = None         # (row, next col) of the current Noop storage row
        self.tape = []            # (op, a, b, c, d) with wire operands as row * num_wires + col
        self.tape_seg = []        # segment of each tape entry: 0 = sequential part, k >= 1 = independent segment k
        self.segment, self._n_segments = 0, 0

    def _w(self, row, col):
        return row * self.nw + col

    def _rec(self, op, a=0, b=0, c=0, d=0):
        self.tape.append((op, a, b, c, d))
        self.tape_seg.append(self.segment)

    def begin_segment(self):
        """ops recorded until end_segment() form one independent unit of witness generation (e.g. one FRI query round): they
        may read what the unsegmented ops and they themselves produced, nothing else -- checked in witness_tape()"""
        self._n_segments += 1
        self.segment = self._n_segments

    def end_segment(self):
        self.segment = 0

    # ---- low level ------------------------------------------------------------------------------------
    def _new_row(self, gtype, param=0, constants=()):
        r = self.cb.add_gate(gtype, param, constants)
        self.rows[r] = {}
        self.trace.update(b"%d,%d,%r;" % (gtype, param, tuple(int(c) for c in constants)))
        return r

    def _set(self, row, col, v):
        self.rows[row][col] = v % P
        return T(row, col, v)

    def _link(self, src, row, col):
        """wire (row, col) carries src's value and is copy-constrained to it"""
        self.rows[row][col] = src.v
        if (src.row, src.col) != (row, col):
            self.cb.connect((src.row, src.col), (row, col))
            self._rec(TAPE_COPY, self._w(row, col), self._w(src.row, src.col))

    def connect(self, a, b):
        assert a.v == b.v, "connect() of targets with different values: %x != %x" % (a.v, b.v)
        self.cb.connect((a.row, a.col), (b.row, b.col))
        self._rec(TAPE_ASSERT_EQ, self._w(a.row, a.col), self._w(b.row, b.col))

    def connect_ext(self, a, b):
        self.connect(a[0], b[0])
        self.connect(a[1], b[1])

    def add_virtual_target(self, v, derived=False):
        """a free witness input (home: a routed wire of a Noop row).  A value carrying `.src` (plonk.Src: its
        position in the caller's flat input vector) is recorded on the tape as an INPUT; `derived` targets are
        recorded by the caller (they are functions of other wires)."""
        if self._free is None or self._free[1] >= self.routed:
            self._free = [self._new_row(GATE_NOOP), 0]
        src = getattr(v, "src", None)
        t = self._set(self._free[0], self._free[1], int(v))
        self._free[1] += 1
        if src is not None:
            self._rec(TAPE_INPUT, self._w(t.row, t.col), src)
        elif not derived:
            self.untagged_inputs = getattr(self, "untagged_inputs", 0) + 1
        return t

    def add_virtual_targets(self, vals):
        return [self.add_virtual_target(v) for v in vals]

    def add_virtual_ext(self, v):
        return (self.add_virtual_target(v[0]), self.add_virtual_target(v[1]))

    def constant(self, v):
        v %= P
        if v not in self.consts:
            key = "const"
            slot = self.open.get(key)
            if slot is None or slot[1] >= 2:
                slot = [self.cb.add_gate(GATE_CONSTANT, 2, constants=[0, 0]), 0]
                self.rows[slot[0]] = {}
                self.trace.update(b"C;")
                self.open[key] = slot
            row, k = slot
            self.trace.update(b"c%d;" % v)        # constants are part of the circuit, not of the witness
            self.cb.rows[row][1][k] = v          # gate constant k of that row
            self.consts[v] = self._set(row, k, v)
            self._rec(TAPE_CONST, self._w(row, k), v)
            slot[1] += 1
        return self.consts[v]

    def constant_ext(self, v):
        return (self.constant(v[0]), self.constant(v[1]))

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def register_public_inputs(self, targets):
        self.public_inputs.extend(targets)

    # ---- ArithmeticGate{20}: out = c0*m0*m1 + c1*addend  (gates/arithmetic.rs) ----------------------------
    def arithmetic(self, c0, m0, m1, c1, addend):
        c0 %= P
        c1 %= P
        key = ("arith", c0, c1)
        slot = self.open.get(key)
        if slot is None or slot[1] >= 20:
            slot = [self._new_row(GATE_ARITHMETIC, 20, constants=[c0, c1]), 0]
            self.open[key] = slot
        row, i = slot
        slot[1] += 1
        self._link(m0, row, 4 * i)
        self._link(m1, row, 4 * i + 1)
        self._link(addend, row, 4 * i + 2)
        self._rec(TAPE_ARITH, self._w(row, 4 * i), c0, c1)
        return self._set(row, 4 * i + 3, c0 * m0.v * m1.v + c1 * addend.v)

    def mul(self, a, b):
        return self.arithmetic(1, a, b, 0, self.zero())

    def add(self, a, b):
        return self.arithmetic(1, a, self.one(), 1, b)

    def sub(self, a, b):
        return self.arithmetic(1, a, self.one(), P - 1, b)

    def mul_const(self, c, a):
        return self.arithmetic(c, a, self.one(), 0, self.zero())

    def mul_add(self, a, b, c):
        return self.arithmetic(1, a, b, 1, c)

    def select(self, bit, x, y):
        """bit ? x : y  =  bit*(x - y) + y"""
        return self.mul_add(bit, self.sub(x, y), y)

    def assert_zero(self, a):
        self.connect(a, self.zero())

    def assert_bool(self, b):
        self.assert_zero(self.arithmetic(1, b, b, P - 1, b))

    # ---- ArithmeticExtensionGate{10}: out = c0*m0*m1 + c1*addend over F_p^2 (gates/arithmetic_extension.rs) ----
    def arithmetic_ext(self, c0, m0, m1, c1, addend):
        c0 %= P
        c1 %= P
        key = ("arith_ext", c0, c1)
        slot = self.open.get(key)
        if slot is None or slot[1] >= 10:
            slot = [self._new_row(GATE_ARITHMETIC_EXT, 10, constants=[c0, c1]), 0]
            self.open[key] = slot
        row, i = slot
        slot[1] += 1
        for k, op in enumerate((m0, m1, addend)):
            self._link(op[0], row, 8 * i + 2 * k)
            self._link(op[1], row, 8 * i + 2 * k + 1)
        pr = emul((m0[0].v, m0[1].v), (m1[0].v, m1[1].v))
        v0 = c0 * pr[0] + c1 * addend[0].v
        v1 = c0 * pr[1] + c1 * addend[1].v
        self._rec(TAPE_ARITH_EXT, self._w(row, 8 * i), c0, c1)
        return (self._set(row, 8 * i + 6, v0), self._set(row, 8 * i + 7, v1))

    def ext_zero(self):
        return (self.zero(), self.zero())

    def ext_one(self):
        return (self.one(), self.zero())

    def ext_mul(self, a, b):
        return self.arithmetic_ext(1, a, b, 0, self.ext_zero())

    def ext_add(self, a, b):
        return self.arithmetic_ext(1, a, self.ext_one(), 1, b)

    def ext_sub(self, a, b):
        return self.arithmetic_ext(1, a, self.ext_one(), P - 1, b)

    def ext_mul_add(self, a, b, c):
        return self.arithmetic_ext(1, a, b, 1, c)

    def ext_mul_sub(self, a, b, c):
        return self.arithmetic_ext(1, a, b, P - 1, c)

    def ext_scalar_mul(self, c, a):
        return self.arithmetic_ext(c, a, self.ext_one(), 0, self.ext_zero())

    def ext_from_base(self, t):
        return (t, self.zero())

    def ext_inverse(self, a):
        """witness the inverse and check a * inv == 1"""
        inv = einv((a[0].v, a[1].v))
        t = (self.add_virtual_target(inv[0], derived=True), self.add_virtual_target(inv[1], derived=True))
        self._rec(TAPE_EXT_INV, self._w(t[0].row, t[0].col), self._w(t[1].row, t[1].col),
                  self._w(a[0].row, a[0].col), self._w(a[1].row, a[1].col))
        prod = self.ext_mul(a, t)
        self.connect_ext(prod, self.ext_one())
        return t

    def ext_div(self, a, b):
        return self.ext_mul(a, self.ext_inverse(b))

    def ext_exp_pow2(self, a, k):
        for _ in range(k):
            a = self.ext_mul(a, a)
        return a

    def ext_exp_const(self, a, e):
        r, cur = None, a
        while e:
            if e & 1:
                r = cur if r is None else self.ext_mul(r, cur)
            e >>= 1
            if e:
                cur = self.ext_mul(cur, cur)
        return r if r is not None else self.ext_one()

    # ---- PoseidonGate (gates/poseidon.rs:329-380) -----------------------------------------------------------
    def permute_swapped(self, inputs, swap=None):
        """12 input targets (+ optional swap bit target) -> 12 output targets; one gate row"""
        row = self._new_row(GATE_POSEIDON)
        sw = swap if swap is not None else self.zero()
        w = poseidon_gate_witness(np.array([t.v for t in inputs], dtype=np.uint64), sw.v)
        for c in range(self.nw):
            self.rows[row][c] = int(w[c])
        for i, t in enumerate(inputs):
            self.cb.connect((t.row, t.col), (row, i))
            self._rec(TAPE_COPY, self._w(row, i), self._w(t.row, t.col))
        self.cb.connect((sw.row, sw.col), (row, 24))
        self._rec(TAPE_COPY, self._w(row, 24), self._w(sw.row, sw.col))
        self._rec(TAPE_POSEIDON, self._w(row, 0))
        return [T(row, 12 + i, int(w[12 + i])) for i in range(12)]

    def hash_n_to_hash_no_pad(self, inputs):
        """overwrite-mode sponge (hasher_chip.rs:122-148): returns 4 targets"""
        z = self.zero()
        state = [z] * 12
        for off in range(0, len(inputs), 8):
            chunk = inputs[off:off + 8]
            state = list(chunk) + state[len(chunk):]
            state = self.permute_swapped(state)
        return state[:4]

    def hash_or_noop(self, inputs):
        if len(inputs) <= 4:
            return list(inputs) + [self.zero()] * (4 - len(inputs))
        return self.hash_n_to_hash_no_pad(inputs)

    # ---- PoseidonMdsGate over the extension algebra (gates/poseidon_mds.rs) --------------------------------------
    def mds_ext(self, state):
        row = self._new_row(GATE_POSEIDON_MDS)
        for i, e in enumerate(state):
            self._link(e[0], row, 2 * i)
            self._link(e[1], row, 2 * i + 1)
        self._rec(TAPE_MDS_EXT, self._w(row, 0))
        out = []
        for r in range(12):
            v0 = sum(CIRC[i] * state[(i + r) % 12][0].v for i in range(12)) + (8 * state[0][0].v if r == 0 else 0)
            v1 = sum(CIRC[i] * state[(i + r) % 12][1].v for i in range(12)) + (8 * state[0][1].v if r == 0 else 0)
            out.append((self._set(row, 2 * (12 + r), v0), self._set(row, 2 * (12 + r) + 1, v1)))
        return out

    # ---- BaseSumGate<2>{32} (gates/base_sum.rs) -----------------------------------------------------------------
    def split_le_32(self, x, n_used=32):
        """x (a target whose value < 2^32) -> its 32 bit targets, little endian (the first n_used are returned)"""
        row = self._new_row(GATE_BASE_SUM, 32)
        self._link(x, row, 0)
        v = x.v
        assert v < (1 << 32)
        self._rec(TAPE_BASE_SUM, self._w(row, 0), 32)
        bits = [self._set(row, 1 + i, (v >> i) & 1) for i in range(32)]
        return bits[:n_used]

    def split_le_64(self, x):
        """all 64 bits of a field element: x = lo + 2^32 * hi with both halves range-checked by BaseSum{32}.
        (Like plonky2's split_le this does not exclude the non-canonical representation x + p < 2^64.)"""
        lo = self.add_virtual_target(x.v & 0xFFFFFFFF, derived=True)
        hi = self.add_virtual_target(x.v >> 32, derived=True)
        self._rec(TAPE_LO32, self._w(lo.row, lo.col), self._w(x.row, x.col))
        self._rec(TAPE_HI32, self._w(hi.row, hi.col), self._w(x.row, x.col))
        bits_lo = self.split_le_32(lo, 32)
        bits_hi = self.split_le_32(hi, 32)
        recomposed = self.arithmetic(1 << 32, hi, self.one(), 1, lo)
        self.connect(recomposed, x)
        return bits_lo + bits_hi

    def le_sum(self, bits):
        acc = self.zero()
        for i in reversed(range(len(bits))):
            acc = self.arithmetic(2, acc, self.one(), 1, bits[i])
        return acc

    # ---- RandomAccessGate{4,4,2} (gates/random_access.rs) ----------------------------------------------------------
    def random_access(self, index_bits_value_target, items):
        """items[16] targets, index target (value < 16) -> items[index]"""
        assert len(items) == 16
        key = "ra"
        slot = self.open.get(key)
        if slot is None or slot[1] >= 4:
            slot = [self._new_row(GATE_RANDOM_ACCESS, RA_PARAM, constants=[0, 0]), 0]
            self.open[key] = slot
            # the two extra-constant wires mirror the gate constants
            self.rows[slot[0]][72] = 0
            self.rows[slot[0]][73] = 0
            for c in range(4):        # unused copies must still satisfy the gate: index 0, list 0, claimed 0, bits 0
                for k in range(18):
                    self.rows[slot[0]][18 * c + k] = 0
                for k in range(4):
                    self.rows[slot[0]][74 + 4 * c + k] = 0
        row, c = slot
        slot[1] += 1
        idx = index_bits_value_target
        assert idx.v < 16
        self._link(idx, row, 18 * c)
        for i, it in enumerate(items):
            self._link(it, row, 18 * c + 2 + i)
        for k in range(4):
            self.rows[row][74 + 4 * c + k] = (idx.v >> k) & 1
        self._rec(TAPE_RANDOM_ACCESS, self._w(row, 0), c)
        return self._set(row, 18 * c + 1, items[idx.v].v)

    # ---- ReducingGate{43} / ReducingExtensionGate{32} (gates/reducing.rs, reducing_extension.rs) ---------------------
    def _reducing(self, ext, alpha, coeffs, old_acc):
        n = 32 if ext else 43
        assert len(coeffs) <= n
        row = self._new_row(GATE_REDUCING_EXT if ext else GATE_REDUCING, n)
        self._link(alpha[0], row, 2)
        self._link(alpha[1], row, 3)
        self._link(old_acc[0], row, 4)
        self._link(old_acc[1], row, 5)
        z = self.zero()
        padded = list(coeffs) + [(z, z) if ext else z] * (n - len(coeffs))
        start_accs = 6 + (2 * n if ext else n)
        acc = (old_acc[0].v, old_acc[1].v)
        al = (alpha[0].v, alpha[1].v)
        out = None
        for i, cf in enumerate(padded):
            if ext:
                self._link(cf[0], row, 6 + 2 * i)
                self._link(cf[1], row, 6 + 2 * i + 1)
            else:
                self._link(cf, row, 6 + i)
        self._rec(TAPE_REDUCING, self._w(row, 0), n, 1 if ext else 0)
        for i, cf in enumerate(padded):
            cv = (cf[0].v, cf[1].v) if ext else (cf.v, 0)
            pr = emul(acc, al)
            acc = ((pr[0] + cv[0]) % P, (pr[1] + cv[1]) % P)
            base = 0 if i == n - 1 else start_accs + 2 * i
            o = (self._set(row, base, acc[0]), self._set(row, base + 1, acc[1]))
            if i == n - 1:
                out = o
        return out

    def reduce_with_powers_base(self, coeffs, alpha):
        """sum_i alpha^i c_i for base-field targets c_i (Horner from the top, 43 per ReducingGate row).
        The gate computes acc <- acc*alpha + c for its coefficients in order, i.e. the FIRST coefficient ends up
        with the highest power: feed the coefficients reversed, highest index first, zero padding at the front."""
        rev = list(reversed(coeffs))
        acc = self.ext_zero()
        # pad at the FRONT of the first chunk so that trailing zero-padding never multiplies the result
        first = len(rev) % 43 or 43
        chunks = [rev[:first]] + [rev[i:i + 43] for i in range(first, len(rev), 43)]
        for ch in chunks:
            z = self.zero()
            acc = self._reducing(False, alpha, [z] * (43 - len(ch)) + ch, acc)
        return acc

    def reduce_with_powers_ext(self, coeffs, alpha):
        rev = list(reversed(coeffs))
        acc = self.ext_zero()
        first = len(rev) % 32 or 32
        chunks = [rev[:first]] + [rev[i:i + 32] for i in range(first, len(rev), 32)]
        for ch in chunks:
            z = self.ext_zero()
            acc = self._reducing(True, alpha, [z] * (32 - len(ch)) + ch, acc)
        return acc

    # ---- finalisation ---------------------------------------------------------------------------------------------
    def finalize_public_inputs(self):
        """hash the registered public inputs in-circuit and pin the digest to a PublicInputGate
        (plonk_verifier_chip.rs:42-53, gates/public_input.rs)."""
        digest = self.hash_n_to_hash_no_pad(list(self.public_inputs))
        row = self._new_row(GATE_PUBLIC_INPUT)
        for i in range(4):
            self._link(digest[i], row, i)
        return [t.v for t in self.public_inputs]

    def structure_hash(self):
        h = self.trace.copy()
        h.update(repr(self.cb.copies).encode())
        return h.hexdigest()

    def sparse_witness(self):
        """(row_idx uint32[k], rows uint64[k][num_wires]) of every non-empty row"""
        idx = sorted(r for r, cols in self.rows.items() if cols)
        vals = np.zeros((len(idx), self.nw), dtype=np.uint64)
        for k, r in enumerate(idx):
            for c, v in self.rows[r].items():
                vals[k, c] = v
        return np.array(idx, dtype=np.uint32), vals

    def witness_tape(self):
        """The recorded witness program with wire operands renumbered to the sparse-row layout of
        sparse_witness(): (tape uint64[n][5], row_idx uint32[k], public-input wire positions int64[n_pi])."""
        idx = sorted(r for r, cols in self.rows.items() if cols)
        rank = np.full(max(idx) + 1, -1, dtype=np.int64)
        rank[idx] = np.arange(len(idx))
        tape = np.array(self.tape, dtype=object)
        ops = tape[:, 0].astype(np.int64)
        out = np.zeros(tape.shape, dtype=np.uint64)
        out[:, 0] = ops.astype(np.uint64)
        masks = np.array([TAPE_WIRE_FIELDS[o] for o in range(14)], dtype=np.int64)[ops]
        for f in range(4):
            col = np.array([int(x) for x in tape[:, 1 + f]], dtype=np.uint64)
            is_wire = (masks >> f) & 1 == 1
            w = col[is_wire].astype(np.int64)
            r = rank[w // self.nw]
            assert (r >= 0).all()
            col[is_wire] = (r * self.nw + w % self.nw).astype(np.uint64)
            out[:, 1 + f] = col
        pi_pos = np.array([rank[t.row] * self.nw + t.col for t in self.public_inputs], dtype=np.int64)
        self.tape_layout = self._segment_layout(out)
        if self.tape_layout is not None:
            order, n_seq, seg_lens = self.tape_layout
            out = out[order]
            self.tape_layout = (n_seq, seg_lens)
        return out, np.array(idx, dtype=np.uint32), pi_pos

    def _segment_layout(self, tape):
        """(permutation, number of sequential entries, lengths of the independent segments) for a tape reordered as
        [CONST entries | unsegmented entries in recording order | segment 1 | segment 2 | ...], or None when some segment reads
        another segment's wires (then the tape stays in recording order and is replayed sequentially)."""
        seg = np.array(self.tape_seg, dtype=np.int64)
        if seg.max(initial=0) == 0:
            return None
        ops = tape[:, 0].astype(np.int64)
        seg = np.where(ops == TAPE_CONST, 0, seg)
        writer = {}

        def reads_writes(e):
            op, a, b, c, d = (int(x) for x in e)
            if op in (TAPE_CONST, TAPE_INPUT):
                return (), (a,)
            if op in (TAPE_COPY, TAPE_LO32, TAPE_HI32):
                return (b,), (a,)
            if op == TAPE_ASSERT_EQ:
                return (a, b), ()
            if op == TAPE_ARITH:
                return (a, a + 1, a + 2), (a + 3,)
            if op == TAPE_ARITH_EXT:
                return tuple(range(a, a + 6)), (a + 6, a + 7)
            if op == TAPE_POSEIDON:
                return tuple(range(a, a + 12)) + (a + 24,), tuple(range(a, a + self.nw))
            if op == TAPE_MDS_EXT:
                return tuple(range(a, a + 24)), tuple(range(a + 24, a + 48))
            if op == TAPE_BASE_SUM:
                return (a,), tuple(range(a + 1, a + 1 + b))
            if op == TAPE_RANDOM_ACCESS:
                return (a + 18 * b,) + tuple(range(a + 18 * b + 2, a + 18 * b + 18)), (a + 18 * b + 1,) + tuple(range(a + 74 + 4 * b, a + 78 + 4 * b))
            if op == TAPE_REDUCING:
                n_c = b * (2 if c else 1)
                return tuple(range(a + 2, a + 6 + n_c)), (a, a + 1) + tuple(range(a + 6 + n_c, a + 6 + n_c + 2 * (b - 1)))
            if op == TAPE_EXT_INV:
                return (c, d), (a, b)
            raise ValueError(op)
        for k in range(tape.shape[0]):
            sk = int(seg[k])
            rd, wr = reads_writes(tape[k])
            for w in rd:
                ws = writer.get(w, 0)
                if ws != 0 and ws != sk:
                    return None
            for w in wr:
                ws = writer.setdefault(w, sk)
                if ws != sk:
                    return None
        is_const = ops == TAPE_CONST
        order = np.concatenate([np.nonzero(is_const)[0], np.nonzero((seg == 0) & ~is_const)[0]] +
                               [np.nonzero(seg == s)[0] for s in range(1, int(seg.max()) + 1)])
        n_seq = int((seg == 0).sum())
        seg_lens = [int((seg == s).sum()) for s in range(1, int(seg.max()) + 1)]
        return order, n_seq, seg_lens
