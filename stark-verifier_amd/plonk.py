"""Host-side prover driver: sequences one plonky2-style proof on top of the gl355 kernels.

This is the host code the reference reaches through `CircuitBuilder::build` / `CircuitData::prove`
(src/plonky2_semaphore/access_set.rs:85-94, recursion.rs:49,167-168, wrapper.rs:36-41,54-55), restated
from the protocol the reference's in-tree verifier pins:
  transcript order      chip/plonk/plonk_verifier_chip.rs:55-154
  oracles / poly order  types/common_data.rs:100-222, types/assigned.rs:26-44
  vanishing terms       chip/plonk/vanishing_poly.rs:18-153 (evaluated on the GPU: csrc/quotient.hip)
  FRI                   chip/fri_chip.rs (batch combine :112-149, fold :168-226, queries :228-327, PoW :364-376)
Every data-parallel stage is a libgl355 call (NTT/LDE/Merkle commit, Z/partial products, quotient,
openings, DEEP quotient, FRI layers, PoW); only the Fiat-Shamir Challenger, the circuit bookkeeping
and witness placement run on the host, as in the reference.
The circuit builder below is this framework's own (gate placement and the circuit digest are not
plonky2's -- its builder is not part of the reference tree), but the gate set, wire layouts,
constraint formulas and proof structure are the reference's.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (GATE_ARITHMETIC, GATE_ARITHMETIC_EXT, GATE_BASE_SUM, GATE_CONSTANT, GATE_MUL_EXT, GATE_NOOP, GATE_POSEIDON,
                   GATE_POSEIDON_MDS, GATE_PUBLIC_INPUT, GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT,
                   Challenger as _CChallenger, Circuit as _CCircuit)
from .api import COSET_SHIFT, P, SALT_SIZE, MerkleTree, PolynomialBatch, _ptr, _u64, deep_batch, eval_polys

UNUSED_SELECTOR = 0xFFFFFFFF


class CircuitConfig:
    """plonky2's CircuitConfig with the values the reference uses (access_set.rs:68-84, recursion.rs:32-48)."""

    def __init__(self, **kw):
        self.num_wires = 135
        self.num_routed_wires = 80
        self.num_constants = 2
        self.num_challenges = 2
        self.max_quotient_degree_factor = 8
        self.zero_knowledge = True
        self.rate_bits = 3
        self.cap_height = 4
        self.proof_of_work_bits = 16
        self.num_query_rounds = 28
        self.reduction_arity_bits = 1      # FriReductionStrategy::ConstantArityBits(1, 5)
        self.final_poly_bits = 5
        self.hasher = 0                    # GenericConfig::Hasher: 0 PoseidonHash, 1 the reference's Bn254PoseidonHash (OuterC)
        self.__dict__.update(kw)

    def fri_reduction_arity_bits(self, degree_bits):
        """ConstantArityBits(a, f): push a while degree_bits > f and degree_bits + rate_bits - a >= cap_height."""
        out, d = [], degree_bits
        a, f = self.reduction_arity_bits, self.final_poly_bits
        while d > f and d + self.rate_bits - a >= self.cap_height:
            out.append(a)
            d -= a
        return out


class Challenger:
    """plonky2::iop::challenger::Challenger on the host (gl355_challenger_*)."""

    def __init__(self, hasher=0):
        self.lib = _lib.load()
        self.c = _CChallenger()
        assert self.lib.gl355_challenger_init_h(C.byref(self.c), hasher) == 0

    def observe(self, elems):
        e = _u64(elems).reshape(-1)
        if e.size:
            self.lib.gl355_challenger_observe(C.byref(self.c), _ptr(e), e.size)

    def squeeze(self, n=1):
        out = np.empty(n, dtype=np.uint64)
        self.lib.gl355_challenger_squeeze(C.byref(self.c), _ptr(out), n)
        return out

    def get_extension_challenge(self):
        return self.squeeze(2)

    def pow_state(self):
        st = np.empty(12, dtype=np.uint64)
        pos = C.c_uint32()
        rc = self.lib.gl355_challenger_pow_state(C.byref(self.c), _ptr(st), C.byref(pos))
        assert rc == 0
        return st, pos.value


def host_hash_no_pad(x, hasher=0):
    lib = _lib.load()
    x = _u64(x).reshape(-1)
    out = np.empty(4, dtype=np.uint64)
    assert lib.gl355_host_hash_no_pad_h(hasher, _ptr(x) if x.size else None, x.size, _ptr(out)) == 0
    return out


def poseidon_gate_witness(inputs, swap):
    lib = _lib.load()
    w = np.empty(135, dtype=np.uint64)
    rc = lib.gl355_poseidon_gate_witness(_ptr(_u64(inputs)), int(swap), _ptr(w))
    assert rc == 0
    return w


class CircuitBuilder:
    """Minimal gate-level builder: rows of gates, copy constraints between routed wires."""

    def __init__(self, config=None, gate_order="own"):
        self.config = config or CircuitConfig()
        # "own": gate indices by decreasing degree (this framework's choice); "plonky2": the order and selector groups upstream's
        # CircuitBuilder::build / selector_polynomials give (by degree, ties by the gate's id string, groups grown greedily from the
        # cheapest gate) -- the tables a plonky2-side exporter would hand over.  Prover, loader and verifier take the
        # indices / groups from the artifact, whatever the order.
        self.gate_order = gate_order
        self.gate_types = []          # [(type, param)] in registration order
        self.rows = []                # [(gate_type_index, [gate constants])]
        self.copies = []              # [((row, col), (row, col))]
        self.num_public_inputs = 0

    def gate_type(self, gtype, param=0):
        key = (gtype, param)
        if key not in self.gate_types:
            self.gate_types.append(key)
        return self.gate_types.index(key)

    def add_gate(self, gtype, param=0, constants=()):
        gi = self.gate_type(gtype, param)
        self.rows.append((gi, list(constants)))
        return len(self.rows) - 1

    def connect(self, a, b):
        assert a[1] < self.config.num_routed_wires and b[1] < self.config.num_routed_wires, "only routed wires can be copied"
        self.copies.append((a, b))

    def build(self, ctx, rng, min_degree_bits=0):
        """Pads (blinding rows + noops), computes selectors / sigmas and commits constants_sigmas on the device."""
        data = self.layout(min_degree_bits)
        cs_values = np.concatenate([data.constants, data.sigmas])
        data.constants_sigmas = PolynomialBatch.from_values(ctx, cs_values, self.config.rate_bits, self.config.cap_height, salt=None,
                                                            hasher=self.config.hasher)
        data.set_digest(data.constants_sigmas.cap)
        return data

    def layout(self, min_degree_bits=0):
        """The host-only part of build(): padding, selector groups, constants and sigma tables, the gl355_circuit
        shape.  No device work; the preprocessed commitment (and the digest that covers it) is added by build()."""
        cfg = self.config
        noop = self.gate_type(GATE_NOOP)
        n_real = len(self.rows)
        # zero-knowledge blinding (plonky2 `CircuitBuilder::blind`): random unconstrained rows hide the wire
        # openings, and pairs of rows whose EVERY routed column carries its own random value, copy-constrained between the two
        # rows of the pair, randomise the Z and partial-product polynomials
        n_blind_wires = n_blind_z = 0
        if cfg.zero_knowledge:
            arities = cfg.fri_reduction_arity_bits(max(min_degree_bits, 13))
            q = cfg.num_query_rounds * (1 + 2 * sum((1 << a) - 1 for a in arities) + 2 * (1 << cfg.final_poly_bits))
            n_blind_wires = 2 + q
            n_blind_z = 2 * (2 * 2 + q)
        blind_start = len(self.rows)
        for _ in range(n_blind_wires + n_blind_z):
            self.add_gate(GATE_NOOP)
        z_pairs = []
        for k in range(n_blind_z // 2):
            r0 = blind_start + n_blind_wires + 2 * k
            for c in range(cfg.num_routed_wires):
                self.connect((r0, c), (r0 + 1, c))
            z_pairs.append((r0, r0 + 1))
        degree_bits = max(min_degree_bits, 2, int(len(self.rows) - 1).bit_length())
        n = 1 << degree_bits
        while len(self.rows) < n:
            self.add_gate(GATE_NOOP)
        # selector groups (plonky2's selector_polynomials): a group of g gates sharing one selector column gets the
        # filter prod_{k != i}(k - s) * (UNUSED - s) of degree g, so every member needs g + degree <= qdf + 1 = 9
        max_deg = cfg.max_quotient_degree_factor + 1
        deg = lambda tp: _GATE_DEGREE[tp[0]](tp[1])
        if self.gate_order == "plonky2":
            order = sorted(range(len(self.gate_types)), key=lambda i: (deg(self.gate_types[i]), gate_id_string(*self.gate_types[i])))
        else:
            order = sorted(range(len(self.gate_types)), key=lambda i: -deg(self.gate_types[i]))
        remap = {old: new for new, old in enumerate(order)}
        gates = [self.gate_types[i] for i in order]
        groups, sel_index = [], []
        start = 0
        if self.gate_order == "plonky2":
            if deg(gates[-1]) + len(gates) - 1 <= max_deg:          # upstream's special case: one selector polynomial for everything
                groups.append((0, len(gates)))
                start = len(gates)
            while start < len(gates):
                size = 0
                while start + size < len(gates) and size + deg(gates[start + size]) < max_deg:
                    size += 1
                assert size >= 1, "a gate of degree >= max_quotient_degree_factor + 1 has no selector group"
                groups.append((start, start + size))
                start += size
        while start < len(gates):
            end = start + 1
            top = deg(gates[start])       # sorted: the first member has the largest degree
            while end < len(gates) and (end - start + 1) + top <= max_deg:
                end += 1
            groups.append((start, end))
            start = end
        for gi in range(len(gates)):
            sel_index.append(next(s for s, (lo, hi) in enumerate(groups) if lo <= gi < hi))
        num_selectors = len(groups)
        consts = np.zeros((num_selectors + cfg.num_constants, n), dtype=np.uint64)
        row_gate = np.zeros(n, dtype=np.int64)
        for r, (gi_old, cs) in enumerate(self.rows):
            gi = remap[gi_old]
            row_gate[r] = gi
            for s in range(num_selectors):
                consts[s, r] = gi if sel_index[gi] == s else UNUSED_SELECTOR
            for j, v in enumerate(cs):
                consts[num_selectors + j, r] = v % P
        # sigma: cyclic permutation inside every copy class, value = k_col * g^row
        routed = cfg.num_routed_wires
        parent = {}

        def find(x):
            while parent.setdefault(x, x) != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for a, b in self.copies:
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[ra] = rb
        classes = {}
        for x in list(parent):
            classes.setdefault(find(x), []).append(x)
        sigma_map = {}
        for members in classes.values():
            members.sort()
            for i, m in enumerate(members):
                sigma_map[m] = members[(i + 1) % len(members)]
        g = pow(7, (P - 1) >> degree_bits, P)
        subgroup = np.empty(n, dtype=object)
        x = 1
        for i in range(n):
            subgroup[i] = x
            x = x * g % P
        k_is = [pow(7, j, P) for j in range(routed)]
        sigmas = np.empty((routed, n), dtype=np.uint64)
        for j in range(routed):
            col = [(k_is[j] * int(s)) % P for s in subgroup]
            sigmas[j] = np.array(col, dtype=np.uint64)
        for (r, c), (r2, c2) in sigma_map.items():
            sigmas[c, r] = (k_is[c2] * int(subgroup[r2])) % P
        data = CircuitData()
        data.config, data.degree_bits, data.gates, data.groups, data.selector_indices = cfg, degree_bits, gates, groups, sel_index
        data.num_selectors = num_selectors
        data.k_is = np.array(k_is, dtype=np.uint64)
        data.sigmas = sigmas
        data.constants = consts
        data.row_gate = row_gate
        data.constants_sigmas = None
        data.copy_classes = [m for m in classes.values() if len(m) > 1]
        data.blind_rows = (blind_start, n_blind_wires, z_pairs, n_real)
        data.num_partial_products = (routed + cfg.max_quotient_degree_factor - 1) // cfg.max_quotient_degree_factor - 1
        data.num_gate_constraints = max([_GATE_CONSTRAINTS[t](p) for t, p in gates] + [0])
        data.fri_arity_bits = cfg.fri_reduction_arity_bits(degree_bits)
        data.circuit_digest = None
        cc = _CCircuit()
        cc.degree_bits, cc.rate_bits = degree_bits, cfg.rate_bits
        cc.num_wires, cc.num_routed_wires, cc.num_constants = cfg.num_wires, routed, cfg.num_constants
        cc.num_selectors, cc.num_challenges = num_selectors, cfg.num_challenges
        cc.max_degree, cc.num_partial_products, cc.num_gates = cfg.max_quotient_degree_factor, data.num_partial_products, len(gates)
        for i, (t, p) in enumerate(gates):
            cc.gates[i].type, cc.gates[i].param, cc.gates[i].selector_index = t, p, sel_index[i]
            cc.gates[i].group_start, cc.gates[i].group_end = groups[sel_index[i]]
        data.c_circuit = cc
        return data


def gate_id_string(gtype, param):
    """`Gate::id()` of the reference's gate set (the strings chip/plonk/gates/mod.rs:141-196 matches on; BaseSumGate{20} is the Semaphore
    circuit's, circuit.rs:42): plonky2 orders gates of equal degree by it"""
    f = "plonky2_field::goldilocks_field::GoldilocksField"
    if gtype == GATE_ARITHMETIC:
        return "ArithmeticGate { num_ops: %d }" % param
    if gtype == GATE_PUBLIC_INPUT:
        return "PublicInputGate"
    if gtype == GATE_NOOP:
        return "NoopGate"
    if gtype == GATE_CONSTANT:
        return "ConstantGate { num_consts: %d }" % param
    if gtype == GATE_BASE_SUM:
        return "BaseSumGate { num_limbs: %d } + Base: 2" % param
    if gtype == GATE_POSEIDON:
        return "PoseidonGate(PhantomData<%s>)<WIDTH=12>" % f
    if gtype == GATE_POSEIDON_MDS:
        return "PoseidonMdsGate(PhantomData<%s>)<WIDTH=12>" % f
    if gtype == GATE_RANDOM_ACCESS:
        return "RandomAccessGate { bits: %d, num_copies: %d, num_extra_constants: %d, _phantom: PhantomData<%s> }<D=2>" % (
            param & 0xFF, (param >> 8) & 0xFF, (param >> 16) & 0xFF, f)
    if gtype == GATE_REDUCING_EXT:
        return "ReducingExtensionGate { num_coeffs: %d }" % param
    if gtype == GATE_REDUCING:
        return "ReducingGate { num_coeffs: %d }" % param
    if gtype == GATE_ARITHMETIC_EXT:
        return "ArithmeticExtensionGate { num_ops: %d }" % param
    if gtype == GATE_MUL_EXT:
        return "MulExtensionGate { num_ops: %d }" % param
    raise ValueError("unknown gate type %r" % (gtype,))


_GATE_DEGREE = {
    GATE_NOOP: lambda p: 0, GATE_CONSTANT: lambda p: 1, GATE_PUBLIC_INPUT: lambda p: 1, GATE_BASE_SUM: lambda p: 2,
    GATE_POSEIDON: lambda p: 7, GATE_ARITHMETIC: lambda p: 3, GATE_ARITHMETIC_EXT: lambda p: 3, GATE_MUL_EXT: lambda p: 3,
    GATE_POSEIDON_MDS: lambda p: 1, GATE_RANDOM_ACCESS: lambda p: (p & 0xFF) + 1, GATE_REDUCING: lambda p: 2,
    GATE_REDUCING_EXT: lambda p: 2,
}
_GATE_CONSTRAINTS = {
    GATE_NOOP: lambda p: 0, GATE_CONSTANT: lambda p: p, GATE_PUBLIC_INPUT: lambda p: 4,
    GATE_BASE_SUM: lambda p: 1 + p, GATE_POSEIDON: lambda p: 123, GATE_ARITHMETIC: lambda p: p,
    GATE_ARITHMETIC_EXT: lambda p: 2 * p, GATE_MUL_EXT: lambda p: 2 * p, GATE_POSEIDON_MDS: lambda p: 24,
    GATE_RANDOM_ACCESS: lambda p: ((p >> 8) & 0xFF) * ((p & 0xFF) + 2) + ((p >> 16) & 0xFF),
    GATE_REDUCING: lambda p: 2 * p, GATE_REDUCING_EXT: lambda p: 2 * p,
}


def key_bytes(seed):
    """the 32-byte blinding key of an integer test seed: its little-endian encoding (tests, tools, reproducible benches).
    Production callers pass None (the library draws the key from the OS CSPRNG) or their own secret 32 bytes."""
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) != 32:
            raise ValueError("a blinding key is 32 bytes")
        return bytes(seed)
    return (int(seed) % (1 << 256)).to_bytes(32, "little")


def blinding_key(seed):
    """seed -> the `const uint8_t* blinding_key` argument of the C ABI: None stays NULL (fresh OS randomness per proof)"""
    if seed is None:
        return None
    return C.cast(C.create_string_buffer(key_bytes(seed), 32), C.c_void_p)


def derive_key(base, index):
    """gl355_derive_key: the per-unit key the batch runtime uses for proof `index` of a batch keyed with `base`"""
    out = C.create_string_buffer(32)
    rc = _lib.load().gl355_derive_key(key_bytes(base), C.c_uint64(int(index)), out)
    if rc != 0:
        raise _lib.Gl355Error(rc, "gl355_derive_key")
    return out.raw


class CircuitData:
    """The prover/verifier data of one built circuit (plonky2 CircuitData / CommonCircuitData)."""

    def set_digest(self, constants_sigmas_cap):
        """circuit digest: hash of the preprocessed commitment and the shape (this framework's own definition;
        plonky2's digest additionally covers its builder's domain separator)"""
        shape = [self.degree_bits, len(self.gates), self.num_selectors] + [(t << 32) | p for t, p in self.gates]
        self.constants_sigmas_cap = np.array(constants_sigmas_cap, dtype=np.uint64).reshape(-1, 4)
        self.circuit_digest = host_hash_no_pad(np.concatenate([np.asarray(constants_sigmas_cap, dtype=np.uint64).reshape(-1),
                                                               np.array(shape, dtype=np.uint64)]), self.config.hasher)

    def prover_data(self, ctx):
        """gl355_prover_data with the sigma values and k_is resident on the device (uploaded once)."""
        key = id(ctx)
        cache = self.__dict__.setdefault("_pd_cache", {})
        if key not in cache:
            lib = ctx.lib
            d_sig, d_k = C.c_void_p(), C.c_void_p()
            ctx.check(lib.gl355_malloc(ctx.h, self.sigmas.nbytes, C.byref(d_sig)))
            ctx.check(lib.gl355_malloc(ctx.h, self.k_is.nbytes, C.byref(d_k)))
            ctx.check(lib.gl355_memcpy_h2d(ctx.h, d_sig, _ptr(self.sigmas), self.sigmas.nbytes))
            ctx.check(lib.gl355_memcpy_h2d(ctx.h, d_k, _ptr(self.k_is), self.k_is.nbytes))
            pd = _lib.ProverData()
            pd.circuit = C.cast(C.pointer(self.c_circuit), C.c_void_p)
            pd.constants_sigmas = self.constants_sigmas.h
            pd.sigmas, pd.k_is = d_sig, d_k
            for i in range(4):
                pd.circuit_digest[i] = int(self.circuit_digest[i])
            cfg = self.config
            pd.cap_height, pd.pow_bits, pd.num_queries = cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds
            pd.n_fri_layers, pd.zero_knowledge = len(self.fri_arity_bits), int(cfg.zero_knowledge)
            pd.hasher = cfg.hasher
            cache[key] = pd
        return cache[key]

    def export_blob(self, row_idx, tape=None, pi_pos=None, n_inputs=0, tape_layout=None, external_digest=None):
        """Serialise the built circuit (+ the sparse witness-row map and optionally a witness tape) as the u64 artifact
        gl355_circuit_load reads (layout in include/gl355.h).  external_digest: 4 words of a circuit digest computed elsewhere (e.g.
        by plonky2's own CircuitBuilder::build, whose digest also covers its domain separator): the artifact is then version 3 and
        carries the expected constants_sigmas cap, which the loader checks against its own commitment of the tables instead of
        re-deriving this framework's digest."""
        cfg = self.config
        cc = self.c_circuit
        hdr = np.zeros(112, dtype=np.uint64)
        hdr[0], hdr[1] = 0x5249433535334c47, 2
        hdr[2:12] = [cc.degree_bits, cc.rate_bits, cc.num_wires, cc.num_routed_wires, cc.num_constants, cc.num_selectors,
                     cc.num_challenges, cc.max_degree, cc.num_partial_products, cc.num_gates]
        for g in range(_lib.MAX_GATES):
            gt = cc.gates[g]
            hdr[12 + 5 * g: 17 + 5 * g] = [gt.type, gt.param, gt.selector_index, gt.group_start, gt.group_end]
        start, n_blind, z_pairs, _ = self.blind_rows
        tape = np.zeros((0, 5), dtype=np.uint64) if tape is None else _u64(tape)
        pi_pos = np.zeros(0, dtype=np.uint64) if pi_pos is None else np.asarray(pi_pos, dtype=np.uint64)
        row_idx = np.asarray(row_idx, dtype=np.uint64)
        hdr[92:106] = [cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds, len(self.fri_arity_bits), int(cfg.zero_knowledge),
                       cfg.hasher, start, n_blind, z_pairs[0][0] if z_pairs else 0, len(z_pairs), row_idx.size, tape.shape[0],
                       n_inputs, pi_pos.size]
        hdr[106:110] = self.circuit_digest if external_digest is None else _u64(external_digest)
        n_seq, seg_lens = tape_layout if tape_layout is not None else (tape.shape[0], [])
        hdr[110], hdr[111] = n_seq, len(seg_lens)
        tail = []
        if external_digest is not None:
            hdr[1] = 3
            tail = [_u64(self.constants_sigmas_cap).reshape(-1)]
        return np.concatenate([hdr, _u64(self.constants).reshape(-1), _u64(self.sigmas).reshape(-1), _u64(self.k_is), row_idx,
                               pi_pos, tape.reshape(-1), np.asarray(seg_lens, dtype=np.uint64)] + tail)

    def verify(self, flat_proof, public_inputs):
        """CircuitData::verify (access_set.rs:170-175) through gl355_verify (host only): raises Gl355Error(GL355_E_VERIFY) with the
        failed check, returns True otherwise"""
        lib = _lib.load()
        cfg = self.config
        vd = _lib.VerifierData()
        cap = np.ascontiguousarray(self.constants_sigmas_cap, dtype=np.uint64)
        kis = np.ascontiguousarray(self.k_is, dtype=np.uint64)
        vd.circuit = C.cast(C.pointer(self.c_circuit), C.c_void_p)
        vd.constants_sigmas_cap, vd.k_is = cap.ctypes.data, kis.ctypes.data
        for i in range(4):
            vd.circuit_digest[i] = int(self.circuit_digest[i])
        vd.cap_height, vd.pow_bits, vd.num_queries = cfg.cap_height, cfg.proof_of_work_bits, cfg.num_query_rounds
        vd.n_fri_layers, vd.zero_knowledge, vd.hasher = len(self.fri_arity_bits), int(cfg.zero_knowledge), int(getattr(cfg, "hasher", 0))
        flat, pi = _u64(flat_proof), _u64(public_inputs)
        rc = lib.gl355_verify(C.byref(vd), _ptr(flat), flat.size, _ptr(pi), pi.size)
        if rc != 0:
            raise _lib.Gl355Error(rc, (lib.gl355_verify_last_error() or b"").decode())
        return True

    def common(self):
        """Plain-dict common data for the verifier restatement in tests/."""
        cfg = self.config
        return dict(degree_bits=self.degree_bits, gates=list(self.gates), groups=list(self.groups),
                    selector_indices=list(self.selector_indices), num_selectors=self.num_selectors,
                    num_constants=cfg.num_constants, num_wires=cfg.num_wires, num_routed_wires=cfg.num_routed_wires,
                    num_challenges=cfg.num_challenges, quotient_degree_factor=cfg.max_quotient_degree_factor,
                    num_partial_products=self.num_partial_products, num_gate_constraints=self.num_gate_constraints,
                    k_is=[int(k) for k in self.k_is], rate_bits=cfg.rate_bits, cap_height=cfg.cap_height,
                    pow_bits=cfg.proof_of_work_bits, num_query_rounds=cfg.num_query_rounds,
                    arity_bits=list(self.fri_arity_bits), hiding=cfg.zero_knowledge, hasher=cfg.hasher,
                    circuit_digest=[int(x) for x in self.circuit_digest],
                    constants_sigmas_cap=self.constants_sigmas_cap.copy())


def fill_blinding(data, wires, rng):
    """Random values on the blinding rows (all wires) and on the Z-blinding pairs (one value per routed column, shared by the
    two rows of a pair: plonky2 `blind`)."""
    start, n_wires_rows, z_pairs, _ = data.blind_rows
    if n_wires_rows:
        wires[:, start:start + n_wires_rows] = rng.integers(0, P, size=(wires.shape[0], n_wires_rows), dtype=np.uint64)
    routed = data.config.num_routed_wires
    for r0, r1 in z_pairs:
        v = rng.integers(0, P, size=routed, dtype=np.uint64)
        wires[:routed, r0] = v
        wires[:routed, r1] = v


def check_copy_constraints(data, wires):
    for members in data.copy_classes:
        vals = {int(wires[c, r]) for r, c in members}
        if len(vals) != 1:
            raise AssertionError("copy constraint violated on %r" % (members[:4],))


def _rand_salt(rng, n_cols):
    return rng.integers(0, P, size=(SALT_SIZE, n_cols), dtype=np.uint64)


class Src(int):
    """an input value that remembers its position in a flat input vector (witness-tape recording)"""
    def __new__(cls, v, src):
        o = int.__new__(cls, int(v))
        o.src = int(src)
        return o


def tag_proof(vals, idxs):
    """zip a parsed proof with the identically parsed index vector: every leaf becomes a Src"""
    if isinstance(vals, dict):
        return {k: tag_proof(v, idxs[k]) for k, v in vals.items()}
    if isinstance(vals, (list, tuple)):
        return type(vals)(tag_proof(v, i) for v, i in zip(vals, idxs))
    if isinstance(vals, np.ndarray):
        return [tag_proof(v, i) for v, i in zip(vals, idxs)]
    return Src(vals, idxs)


def parse_proof_tagged(data, flat, public_inputs, offset=0):
    """parse_proof whose leaves are Src(value, offset + position in flat ‖ public_inputs)"""
    flat = _u64(flat)
    idx = np.arange(offset, offset + flat.size, dtype=np.uint64)
    idx[:8] = flat[:8]
    proof = tag_proof(parse_proof(data, flat), parse_proof(data, idx))
    proof["public_inputs"] = [Src(v, offset + flat.size + i) for i, v in enumerate(_u64(public_inputs))]
    return proof


def parse_proof(data, flat):
    """Flat u64 proof of gl355_prove (layout in include/gl355.h) -> the dict the verifier restatement reads.
    data: CircuitData, or its common() dict."""
    if isinstance(data, dict):
        rate_bits, n_const, n_routed, n_wires = data["rate_bits"], data["num_selectors"] + data["num_constants"], data["num_routed_wires"], data["num_wires"]
        n_pp, qdf = data["num_partial_products"], data["quotient_degree_factor"]
    else:
        cfg = data.config
        rate_bits, n_const, n_routed, n_wires = cfg.rate_bits, data.num_selectors + cfg.num_constants, cfg.num_routed_wires, cfg.num_wires
        n_pp, qdf = data.num_partial_products, cfg.max_quotient_degree_factor
    total, degree_bits, n_layers, nq, n_pi, zk, cap_h, nch = [int(v) for v in flat[:8]]
    n_cap = 1 << cap_h
    lde_bits = degree_bits + rate_bits
    n = 1 << degree_bits
    pos = [8]

    def take(k, shape=None):
        v = flat[pos[0]:pos[0] + k]
        pos[0] += k
        return v.reshape(shape) if shape else v
    wires_cap, zs_cap, q_cap = take(4 * n_cap, (n_cap, 4)), take(4 * n_cap, (n_cap, 4)), take(4 * n_cap, (n_cap, 4))
    counts = [("constants", n_const), ("plonk_sigmas", n_routed), ("wires", n_wires), ("plonk_zs", nch),
              ("partial_products", nch * n_pp), ("quotient_polys", nch * qdf), ("plonk_zs_next", nch)]
    openings = {name: take(2 * k, (k, 2)) for name, k in counts}
    caps = take(n_layers * n_cap * 4, (n_layers, n_cap, 4))
    final_poly = take(2 * (n >> n_layers), (-1, 2))
    pow_witness = int(take(1)[0])
    widths = [n_const + n_routed, n_wires, nch * (1 + n_pp), nch * qdf]
    depth0 = lde_bits - cap_h
    queries = []
    for _ in range(nq):
        x_index = int(take(1)[0])
        initial = []
        for o in range(4):
            ll = widths[o] + (SALT_SIZE if (zk and o > 0) else 0)
            initial.append((take(ll), take(depth0 * 4, (depth0, 4))))
        steps = []
        for l in range(n_layers):
            d = lde_bits - 1 - l - cap_h
            steps.append((take(4), take(d * 4, (d, 4))))
        queries.append(dict(index=x_index, initial_trees=initial, steps=steps))
    assert pos[0] == total == flat.size
    return dict(wires_cap=wires_cap, plonk_zs_partial_products_cap=zs_cap, quotient_polys_cap=q_cap, openings=openings,
                opening_proof=dict(commit_phase_merkle_caps=[caps[l] for l in range(n_layers)], query_round_proofs=queries,
                                   final_poly=final_poly, pow_witness=pow_witness))


class NativeCircuit:
    """A circuit artifact loaded into the library (gl355_circuit_load): proving needs no host-side circuit data any more.
    One instance serves every Context of its device."""

    def __init__(self, ctx, blob):
        self.ctx = ctx
        self.blob = np.ascontiguousarray(blob, dtype=np.uint64)
        self.h = C.c_void_p()
        ctx.check(ctx.lib.gl355_circuit_load(ctx.h, _ptr(self.blob), self.blob.size, C.byref(self.h)))
        pw, npi, nrows, nin, db = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint32()
        ctx.lib.gl355_circuit_info(self.h, C.byref(pw), C.byref(npi), C.byref(nrows), C.byref(nin), C.byref(db))
        self.proof_words, self.n_public_inputs, self.n_rows, self.n_inputs, self.degree_bits = pw.value, npi.value, nrows.value, nin.value, db.value

    def prove_rows(self, ctx, rows, public_inputs, seed):
        rows, pi = _u64(rows), _u64(public_inputs)
        flat = np.empty(self.proof_words, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_circuit_prove_rows(ctx.h, self.h, _ptr(rows), _ptr(pi), pi.size, blinding_key(seed), _ptr(flat), flat.size))
        return flat

    def prove_tape(self, ctx, inputs, seed):
        """-> (flat proof, public inputs); raises Gl355Error (GL355_E_WITNESS) when the inputs do not satisfy the circuit"""
        inputs = _u64(inputs)
        flat = np.empty(self.proof_words, dtype=np.uint64)
        pis = np.empty(self.n_public_inputs, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_circuit_prove_tape(ctx.h, self.h, _ptr(inputs), inputs.size, blinding_key(seed), _ptr(flat), flat.size,
                                                   _ptr(pis)))
        return flat, pis

    def prove_tape_units(self, ctx, inputs, seeds):
        """gl355_circuit_prove_tape_units: inputs [units][n_inputs] (each unit: its inner proofs' flat words | public inputs) proven in
        lock-step on `ctx`; seeds: one blinding key per unit, or None.  -> (proofs [units][words], public inputs [units][n_pi])"""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, self.n_inputs)
        units = inputs.shape[0]
        flat = np.empty((units, self.proof_words), dtype=np.uint64)
        pis = np.empty((units, self.n_public_inputs), dtype=np.uint64)
        keys = None if seeds is None else C.cast(C.create_string_buffer(b"".join(key_bytes(s) for s in seeds), 32 * units), C.c_void_p)
        ctx.check(ctx.lib.gl355_circuit_prove_tape_units(ctx.h, self.h, units, _ptr(inputs), self.n_inputs, keys, _ptr(flat), _ptr(pis)))
        return flat, pis

    def witness_rows(self, ctx, inputs, on_device):
        """gl355_circuit_witness_rows: (rows [units][n_rows][num_wires], public inputs [units][n_pi]) of `inputs` [units][n_inputs]
        by the host replay or the device tape interpreter; raises Gl355Error(GL355_E_WITNESS) with .failed_entry on invalid inputs"""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, self.n_inputs)
        units = inputs.shape[0]
        rows = np.empty((units, self.n_rows, 135), dtype=np.uint64)
        pis = np.empty((units, self.n_public_inputs), dtype=np.uint64)
        failed = C.c_uint64(0)
        rc = ctx.lib.gl355_circuit_witness_rows(ctx.h, self.h, units, _ptr(inputs), self.n_inputs, int(on_device), _ptr(rows), _ptr(pis), C.byref(failed))
        if rc != 0:
            err = _lib.Gl355Error(rc, (ctx.lib.gl355_last_error(ctx.h) or b"").decode())
            err.failed_entry = failed.value
            raise err
        return rows, pis

    def semaphore_prove(self, ctx, private_key, topic, index, siblings, seed):
        sk, tp, sib = _u64(private_key), _u64(topic), _u64(siblings)
        flat = np.empty(self.proof_words, dtype=np.uint64)
        pis = np.empty(12, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_semaphore_prove(ctx.h, self.h, _ptr(sk), _ptr(tp), int(index), _ptr(sib), sib.shape[0], blinding_key(seed),
                                                _ptr(flat), flat.size, _ptr(pis)))
        return flat, pis

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):          # see PolynomialBatch.close
                self.ctx.lib.gl355_circuit_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def semaphore_units(ctxs, sem, rec, private_keys, topic, tree_digests, member_indices, seed_base, want_proofs=False):
    """gl355_semaphore_units: the native batch runtime (recursion.rs:300-308 `par_iter` of make_signal + the verification
    circuit per signal): one host thread per context inside the library, unit j on context j mod len(ctxs).
    -> leaves [count][8] (nullifier | topic), proofs [count][words] or None, units proven per context"""
    lib = ctxs[0].lib
    hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    sk, tp, dg = _u64(private_keys), _u64(topic), _u64(tree_digests)
    idx = np.ascontiguousarray(member_indices, dtype=np.uint64)
    leaves = np.zeros((idx.size, 8), dtype=np.uint64)
    words = (rec or sem).proof_words
    proofs = np.empty((idx.size, words), dtype=np.uint64) if want_proofs else None
    per = (C.c_uint32 * len(ctxs))()
    rc = lib.gl355_semaphore_units(hs, len(ctxs), sem.h, rec.h if rec is not None else None, _ptr(sk), sk.shape[0], _ptr(tp), _ptr(dg),
                                   _ptr(idx), idx.size, blinding_key(seed_base), _ptr(leaves), _ptr(proofs) if want_proofs else None, per)
    if rc != 0:
        msgs = [(c.lib.gl355_last_error(c.h) or b"").decode() for c in ctxs]
        raise _lib.Gl355Error(rc, "; ".join(m for m in msgs if m))
    return leaves, proofs, list(per)


def prove(ctx, data, wires, public_inputs, seed, flat_only=False):
    """CircuitData::prove through the single resident C entry point gl355_prove (csrc/prover.hip)."""
    lib = ctx.lib
    cfg = data.config
    pd = data.prover_data(ctx)
    pi = _u64(public_inputs)
    words = lib.gl355_proof_words(C.byref(pd))
    flat = np.empty(words, dtype=np.uint64)
    w = wires if hasattr(wires, "data_ptr") else np.ascontiguousarray(wires, dtype=np.uint64)
    ctx.check(lib.gl355_prove(ctx.h, C.byref(pd), _ptr(w), _ptr(pi), pi.size, blinding_key(seed), _ptr(flat), words))
    if flat_only:
        return flat
    proof = parse_proof(data, flat)
    proof["public_inputs"] = pi.copy()
    return proof


def prove_sparse(ctx, data, row_idx, rows, public_inputs, seed, flat_only=False):
    """gl355_prove_sparse: witness = its non-trivial rows; blinding rows are generated on the device."""
    lib = ctx.lib
    pd = data.prover_data(ctx)
    pi = _u64(public_inputs)
    words = lib.gl355_proof_words(C.byref(pd))
    flat = np.empty(words, dtype=np.uint64)
    idx = np.ascontiguousarray(row_idx, dtype=np.uint32)
    rows = _u64(rows)
    start, n_blind, z_pairs, _ = data.blind_rows
    z_start = z_pairs[0][0] if z_pairs else 0
    ctx.check(lib.gl355_prove_sparse(ctx.h, C.byref(pd), idx.ctypes.data_as(C.c_void_p), _ptr(rows), idx.size, start, n_blind,
                                     z_start, len(z_pairs), _ptr(pi), pi.size, blinding_key(seed), _ptr(flat), words))
    if flat_only:
        return flat
    proof = parse_proof(data, flat)
    proof["public_inputs"] = pi.copy()
    return proof


def prove_sparse_units(ctx, data, row_idx, rows, public_inputs, seeds):
    """gl355_prove_sparse_units: len(rows) independent witnesses of one circuit proven in lock-step on `ctx`; seeds: one blinding
    key (int / 32 bytes) per unit, or None for OS randomness.  -> flat proofs [units][words]"""
    lib = ctx.lib
    pd = data.prover_data(ctx)
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    pis = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(rows.shape[0], -1)
    units = rows.shape[0]
    words = lib.gl355_proof_words(C.byref(pd))
    flat = np.empty((units, words), dtype=np.uint64)
    idx = np.ascontiguousarray(row_idx, dtype=np.uint32)
    start, n_blind, z_pairs, _ = data.blind_rows
    z_start = z_pairs[0][0] if z_pairs else 0
    keys = None if seeds is None else C.cast(C.create_string_buffer(b"".join(key_bytes(s) for s in seeds), 32 * units), C.c_void_p)
    ctx.check(lib.gl355_prove_sparse_units(ctx.h, C.byref(pd), units, idx.ctypes.data_as(C.c_void_p), _ptr(rows), idx.size, start, n_blind,
                                           z_start, len(z_pairs), _ptr(pis), pis.shape[1], keys, _ptr(flat), words))
    return flat


def prove_staged(ctx, data, wires, public_inputs, rng, timings=None):
    """The same proof sequenced stage by stage from Python over the individual C-ABI entry points (used for
    per-stage timings and to cross-check gl355_prove); wires[num_wires][n] is the full witness."""
    import time
    cfg = data.config
    lib = ctx.lib
    n = 1 << data.degree_bits
    N = n << cfg.rate_bits
    nch = cfg.num_challenges
    salted = cfg.zero_knowledge
    t_last = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            ctx.sync()
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - t_last[0])
            t_last[0] = now

    pi = _u64(public_inputs)
    pi_hash = host_hash_no_pad(pi)
    ch = Challenger()
    ch.observe(data.circuit_digest)
    ch.observe(pi_hash)
    # ---- wires ---------------------------------------------------------------------------------------
    wires_batch = PolynomialBatch.from_values(ctx, wires, cfg.rate_bits, cfg.cap_height, salt=_rand_salt(rng, N) if salted else None)
    wires_cap = wires_batch.cap
    ch.observe(wires_cap)
    betas, gammas = ch.squeeze(nch), ch.squeeze(nch)
    lap("wires commit")
    # ---- Z / partial products ---------------------------------------------------------------------------
    routed = cfg.num_routed_wires
    zs, pps = [], []
    for c in range(nch):
        z, pp = ctx.zs_partial_products(wires[:routed], data.sigmas, data.k_is, cfg.max_quotient_degree_factor, int(betas[c]), int(gammas[c]))
        zs.append(z)
        pps.append(pp)
    zs_pp_values = np.concatenate([np.stack(zs)] + pps)
    zs_batch = PolynomialBatch.from_values(ctx, zs_pp_values, cfg.rate_bits, cfg.cap_height, salt=_rand_salt(rng, N) if salted else None)
    zs_cap = zs_batch.cap
    ch.observe(zs_cap)
    alphas = ch.squeeze(nch)
    lap("Z/partial products + commit")
    # ---- quotient -----------------------------------------------------------------------------------------
    qdf = cfg.max_quotient_degree_factor
    quot_coeffs = np.empty((nch * qdf, n), dtype=np.uint64)
    ctx.check(lib.gl355_quotient(ctx.h, C.byref(data.c_circuit), data.constants_sigmas.h, wires_batch.h, zs_batch.h,
                                 _ptr(data.k_is), _ptr(betas), _ptr(gammas), _ptr(alphas), _ptr(pi_hash), _ptr(quot_coeffs)))
    quot_batch = PolynomialBatch.from_coeffs(ctx, quot_coeffs, cfg.rate_bits, cfg.cap_height, salt=_rand_salt(rng, N) if salted else None)
    quot_cap = quot_batch.cap
    ch.observe(quot_cap)
    zeta = ch.get_extension_challenge()
    lap("quotient + commit")
    # ---- openings -------------------------------------------------------------------------------------------
    g = pow(7, (P - 1) >> data.degree_bits, P)
    zeta_next = np.array([int(zeta[0]) * g % P, int(zeta[1]) * g % P], dtype=np.uint64)
    oracles = [data.constants_sigmas, wires_batch, zs_batch, quot_batch]
    all_polys = [(o, i) for o in oracles for i in range(o.batch)]
    zs_polys = [(zs_batch, i) for i in range(nch)]
    ev = eval_polys(ctx, all_polys, zeta)
    ev_next = eval_polys(ctx, zs_polys, zeta_next)
    n_cs = data.constants_sigmas.batch
    n_const = data.num_selectors + cfg.num_constants
    openings = dict(constants=ev[:n_const], plonk_sigmas=ev[n_const:n_cs], wires=ev[n_cs:n_cs + cfg.num_wires],
                    plonk_zs=ev[n_cs + cfg.num_wires:n_cs + cfg.num_wires + nch],
                    partial_products=ev[n_cs + cfg.num_wires + nch:n_cs + cfg.num_wires + zs_batch.batch],
                    quotient_polys=ev[n_cs + cfg.num_wires + zs_batch.batch:], plonk_zs_next=ev_next)
    ch.observe(ev)
    ch.observe(ev_next)
    fri_alpha = ch.get_extension_challenge()
    lap("openings")
    # ---- DEEP quotient (prove_openings) ------------------------------------------------------------------------
    acc = np.zeros(2 * n, dtype=np.uint64)
    acc = deep_batch(ctx, all_polys, fri_alpha, zeta, acc)
    acc = deep_batch(ctx, zs_polys, fri_alpha, zeta_next, acc)
    lap("DEEP quotient")
    # ---- FRI: commit phase, proof of work, layer openings (one resident call: csrc/prover.hip) ---------------------
    arity = np.array(data.fri_arity_bits, dtype=np.uint32)
    n_layers = arity.size
    lde_bits = data.degree_bits + cfg.rate_bits
    n_cap = 1 << cfg.cap_height
    nq = cfg.num_query_rounds
    depths = [lde_bits - 1 - l - cfg.cap_height for l in range(n_layers)]
    caps = np.empty((n_layers, n_cap, 4), dtype=np.uint64)
    final_poly = np.empty((n >> n_layers, 2), dtype=np.uint64)
    pow_witness = C.c_uint64()
    x_indices = np.empty(nq, dtype=np.uint64)
    step_evals = np.empty((nq, n_layers, 4), dtype=np.uint64)
    step_sibs = np.empty((nq, max(1, sum(depths)), 4), dtype=np.uint64)
    ctx.check(lib.gl355_fri_prove(ctx.h, _ptr(acc), data.degree_bits, cfg.rate_bits, cfg.cap_height,
                                  arity.ctypes.data_as(C.c_void_p), n_layers, cfg.proof_of_work_bits, nq, C.byref(ch.c),
                                  _ptr(caps), _ptr(final_poly), C.byref(pow_witness), _ptr(x_indices), _ptr(step_evals),
                                  _ptr(step_sibs)))
    lap("FRI commit + PoW + layer openings")
    # ---- initial-tree openings (fri_prover_query_round) -----------------------------------------------------------------
    opened = [o.open_batch(x_indices) for o in oracles]
    queries = []
    offs = np.concatenate([[0], np.cumsum(depths)]).astype(int)
    for qi in range(nq):
        steps = [(step_evals[qi, l], step_sibs[qi, offs[l]:offs[l + 1]]) for l in range(n_layers)]
        queries.append(dict(index=int(x_indices[qi]), initial_trees=[opened[o][qi] for o in range(len(oracles))], steps=steps))
    lap("initial-tree openings")
    trees_caps = [caps[l] for l in range(n_layers)]
    pow_witness = pow_witness.value
    proof = dict(wires_cap=wires_cap, plonk_zs_partial_products_cap=zs_cap, quotient_polys_cap=quot_cap,
                 openings=openings,
                 opening_proof=dict(commit_phase_merkle_caps=trees_caps, query_round_proofs=queries,
                                    final_poly=final_poly, pow_witness=pow_witness),
                 public_inputs=pi.copy())
    for o in (wires_batch, zs_batch, quot_batch):
        o.close()
    return proof
