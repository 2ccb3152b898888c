"""Semaphore access set, circuit and signal -- host mirror of src/plonky2_semaphore/{access_set,circuit,signal}.rs.

Same statement as the reference: 12 public inputs (merkle_root | nullifier | topic, circuit.rs:27-32),
a depth-h Merkle membership proof of public_key = Poseidon(private_key | 0^4) with the index bits
taken from a base-2 split (circuit.rs:42-51), and nullifier = Poseidon(private_key | topic)
(circuit.rs:52-57).  Gate placement is this framework's own builder (plonk.CircuitBuilder).
"""
import numpy as np

from ._lib import GATE_BASE_SUM, GATE_CONSTANT, GATE_POSEIDON, GATE_PUBLIC_INPUT
from .api import MerkleTree
import ctypes as C

from . import _lib
from .api import _ptr
from .plonk import (CircuitBuilder, CircuitConfig, check_copy_constraints, fill_blinding, host_hash_no_pad,
                    poseidon_gate_witness, prove, prove_sparse, prove_staged)

IN, OUT, SWAP = 0, 12, 24   # PoseidonGate wire offsets (chip/plonk/gates/poseidon.rs:329-345)


def semaphore_circuit(builder, h):
    """circuit.rs:25-65 for a tree of height h: gate rows + copy constraints; returns the row map"""
    b = builder
    r = dict()
    r["pi"] = b.add_gate(GATE_PUBLIC_INPUT)
    r["h1"] = b.add_gate(GATE_POSEIDON)      # public-input hash, permutation 1 (inputs pi[0..8])
    r["h2"] = b.add_gate(GATE_POSEIDON)      # permutation 2 (pi[8..12] overwrite lanes 0..3)
    r["bits"] = b.add_gate(GATE_BASE_SUM, h)  # split_le(public_key_index, h)
    r["leaf"] = b.add_gate(GATE_POSEIDON)    # public key = H(private_key | 0^4)
    r["m"] = [b.add_gate(GATE_POSEIDON) for _ in range(h)]
    r["null"] = b.add_gate(GATE_POSEIDON)    # nullifier = H(private_key | topic)
    r["zero"] = b.add_gate(GATE_CONSTANT, 2, constants=(0, 0))
    zero = (r["zero"], 0)
    # public-input hash chain and PublicInputGate
    for j in range(8, 12):
        b.connect((r["h1"], IN + j), zero)
    for j in range(4, 12):
        b.connect((r["h1"], OUT + j), (r["h2"], IN + j))
    for j in range(4):
        b.connect((r["h2"], OUT + j), (r["pi"], j))
    for row in (r["h1"], r["h2"], r["leaf"], r["null"]):
        b.connect((row, SWAP), zero)
    # leaf hash
    for j in range(4, 12):
        b.connect((r["leaf"], IN + j), zero)
    # Merkle path: state in lanes 0..3, sibling in 4..7, swap = index bit (merkle_proof_chip.rs:58-70)
    prev = r["leaf"]
    for i, row in enumerate(r["m"]):
        for j in range(4):
            b.connect((row, IN + j), (prev, OUT + j))
        for j in range(8, 12):
            b.connect((row, IN + j), zero)
        b.connect((row, SWAP), (r["bits"], 1 + i))
        prev = row
    # public inputs: root | nullifier | topic
    for j in range(4):
        b.connect((r["h1"], IN + j), (prev, OUT + j))                 # merkle_root
        b.connect((r["h1"], IN + 4 + j), (r["null"], OUT + j))        # nullifier
        b.connect((r["h2"], IN + j), (r["null"], IN + 4 + j))         # topic
        b.connect((r["null"], IN + j), (r["leaf"], IN + j))           # private key
    for j in range(8, 12):
        b.connect((r["null"], IN + j), zero)
    return r


class Signal:
    """signal.rs:11-15"""

    def __init__(self, topics, nullifier, proof):
        self.topics, self.nullifier, self.proof = topics, nullifier, proof


class AccessSet:
    """access_set.rs:25 -- AccessSet(pub MerkleTree<F, PoseidonHash>) over the members' public keys."""

    def __init__(self, ctx, public_keys):
        self.ctx = ctx
        self.tree = MerkleTree(ctx, np.ascontiguousarray(public_keys, dtype=np.uint64), 0)
        self._circuit = None

    @staticmethod
    def public_key(private_key):
        """signal.rs:32-39: hash_no_pad(private_key | 0^4)"""
        return host_hash_no_pad(np.concatenate([np.asarray(private_key, dtype=np.uint64), np.zeros(4, np.uint64)]))

    def tree_height(self):
        return int(self.tree.leaves.shape[0]).bit_length() - 1

    # ---- circuit.rs:25-65 ------------------------------------------------------------------------------
    def semaphore_circuit(self, builder):
        return semaphore_circuit(builder, self.tree_height())

    # ---- circuit.rs:67-99 ---------------------------------------------------------------------------------
    def fill_semaphore_targets(self, data, rows, private_key, topic, public_key_index, rng):
        cfg = data.config
        n = 1 << data.degree_bits
        h = self.tree_height()
        wires = np.zeros((cfg.num_wires, n), dtype=np.uint64)
        sk = np.asarray(private_key, dtype=np.uint64)
        tp = np.asarray(topic, dtype=np.uint64)
        z4 = np.zeros(4, np.uint64)
        siblings = self.tree.prove(public_key_index)
        wires[:, rows["leaf"]] = poseidon_gate_witness(np.concatenate([sk, z4, z4]), 0)
        state = wires[OUT:OUT + 4, rows["leaf"]].copy()
        wires[0, rows["bits"]] = public_key_index
        for i, row in enumerate(rows["m"]):
            bit = (public_key_index >> i) & 1
            wires[1 + i, rows["bits"]] = bit
            wires[:, row] = poseidon_gate_witness(np.concatenate([state, siblings[i], z4]), bit)
            state = wires[OUT:OUT + 4, row].copy()
        root = state
        wires[:, rows["null"]] = poseidon_gate_witness(np.concatenate([sk, tp, z4]), 0)
        nullifier = wires[OUT:OUT + 4, rows["null"]].copy()
        public_inputs = np.concatenate([root, nullifier, tp])
        wires[:, rows["h1"]] = poseidon_gate_witness(np.concatenate([public_inputs[:8], z4]), 0)
        st1 = wires[OUT:OUT + 12, rows["h1"]].copy()
        wires[:, rows["h2"]] = poseidon_gate_witness(np.concatenate([public_inputs[8:], st1[4:]]), 0)
        wires[0:4, rows["pi"]] = wires[OUT:OUT + 4, rows["h2"]]
        fill_blinding(data, wires, rng)
        return wires, public_inputs

    def witness_rows(self, rows, private_key, topic, public_key_index):
        """the same witness as fill_semaphore_targets as sparse rows, computed by one C call
        (gl355_semaphore_witness); returns (row_idx, rows[h+7][135], public_inputs)."""
        h = self.tree_height()
        lib = _lib.load()
        sib = np.ascontiguousarray(self.tree.prove(public_key_index), dtype=np.uint64)
        vals = np.empty((h + 7, 135), dtype=np.uint64)
        pi = np.empty(12, dtype=np.uint64)
        sk = np.ascontiguousarray(private_key, dtype=np.uint64)
        tp = np.ascontiguousarray(topic, dtype=np.uint64)
        rc = lib.gl355_semaphore_witness(_ptr(sk), _ptr(tp), int(public_key_index), _ptr(sib), h, _ptr(vals), _ptr(pi))
        assert rc == 0
        idx = np.array([rows["pi"], rows["h1"], rows["h2"], rows["bits"], rows["leaf"]] + list(rows["m"]) + [rows["null"], rows["zero"]],
                       dtype=np.uint32)
        return idx, vals, pi

    def build(self, rng, config=None, gate_order="own"):
        if self._circuit is None:
            builder = CircuitBuilder(config or CircuitConfig(), gate_order=gate_order)
            rows = self.semaphore_circuit(builder)
            data = builder.build(self.ctx, rng)
            self._circuit = (data, rows)
        return self._circuit

    # ---- access_set.rs:61-104 ---------------------------------------------------------------------------------
    def make_signal_fast(self, private_key, topic, public_key_index, seed, flat_only=False):
        """make_signal with the sparse witness path: 2 C calls per proof (witness rows, gl355_prove_sparse)."""
        data, rows = self.build(None)
        idx, vals, public_inputs = self.witness_rows(rows, private_key, topic, public_key_index)
        proof = prove_sparse(self.ctx, data, idx, vals, public_inputs, seed, flat_only=flat_only)
        return Signal([np.asarray(topic, np.uint64)], [public_inputs[4:8].copy()], proof), data

    def make_signal(self, private_key, topic, public_key_index, rng, timings=None, check=False, staged=False):
        data, rows = self.build(rng)
        wires, public_inputs = self.fill_semaphore_targets(data, rows, private_key, topic, public_key_index, rng)
        assert np.array_equal(public_inputs[:4], self.tree.cap[0]), "witness root != access-set root"
        if check:
            check_copy_constraints(data, wires)
        if staged or timings is not None:
            proof = prove_staged(self.ctx, data, wires, public_inputs, rng, timings)
        else:
            proof = prove(self.ctx, data, wires, public_inputs, int(rng.integers(0, 1 << 62)))
        nullifier = host_hash_no_pad(np.concatenate([np.asarray(private_key, np.uint64), np.asarray(topic, np.uint64)]))
        assert np.array_equal(nullifier, public_inputs[4:8])
        return Signal([np.asarray(topic, np.uint64)], [nullifier], proof), data
