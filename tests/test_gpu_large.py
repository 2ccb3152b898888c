"""GPU parity at the sizes BASELINE.json states (round-2 additions):
  cfg-3  Poseidon Merkle trees over 2^22 leaves (FRI-layer shape L=4 / cap 4 in full against the oracle; wires-like L=135 in full as well
         (every digest and the cap); the reference's group tree 2^20 x 4 / cap 0, signal.rs:40, in full)
  cfg-2  NTT / LDE at 2^21..2^23 (the two-level-table branch of ntt.hip, never reached below 2^21)
  cfg-4  one depth-20 unit (2^20-member access set, signer 12, signal.rs:42) through the native batch runtime: Semaphore proof and
         recursive proof byte-identical to the CPU restatement of prove() and to the committed digests tests/golden/unit_depth20.json
  a fan-in-2 aggregation node (recursion.rs:25-185, n = 2^15) byte-identical GPU == CPU
  the blinding stream kernel (ChaCha20) against the oracle's restatement
All through the C ABI, tolerance zero."""
import ctypes as C
import importlib
import json
import os

import numpy as np
import pytest

import cpu_semaphore as cs
import cpu_unit as cu
from oracle_lib import P, key_bytes, rand_field

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def eq(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError("mismatch at %d/%d positions, first %s" % (len(bad), a.size, bad[0]))


# ---- cfg-3 -----------------------------------------------------------------------------------------------------------------
def test_merkle_2p22_fri_layer_shape_full(gl, ctx, orc):
    rng = np.random.default_rng(0x356)
    leaves = rand_field(rng, (1 << 22, 4))
    t = gl.MerkleTree(ctx, leaves, 4)
    dig, cap = orc.merkle_build(leaves, 4)
    eq(t.cap, cap)
    eq(t.digests, dig)
    for i in (0, 1, (1 << 22) - 1, 0x2AAAAA, 12):
        assert orc.merkle_verify(leaves[i], i, t.prove(i), t.cap, 4)


def test_group_tree_2p20_cap0_full(gl, ctx, orc):
    """MerkleTree::new(public keys, 0) of signal.rs:40 / access_set.rs:205 at the depth-20 size"""
    rng = np.random.default_rng(0x357)
    leaves = rand_field(rng, (1 << 20, 4))
    t = gl.MerkleTree(ctx, leaves, 0)
    dig, cap = orc.merkle_build(leaves, 0)
    eq(t.cap, cap)
    eq(t.digests, dig)


def test_merkle_2p22_wires_shape_full(gl, ctx, orc):
    """2^22 leaves x 135 (4.5 GB of leaves, generated on the device): 75.5 M permutations.  EVERY digest of the plonky2-layout buffer
    and the whole cap against the oracle's MerkleTree::new over the same leaves (OpenMP over the host cores), plus paths through
    gl355_merkle_prove."""
    import torch
    n, L, cap_h = 1 << 22, 135, 4
    g = torch.Generator(device="cuda")
    g.manual_seed(0x356)
    leaves = torch.randint(0, (1 << 63) - 1, (n, L), dtype=torch.int64, device="cuda", generator=g)   # canonical (< p) by construction
    n_dig = 2 * (n - (1 << cap_h))
    dig = torch.empty((n_dig, 4), dtype=torch.int64, device="cuda")
    cap = torch.empty((1 << cap_h, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.check(ctx.lib.gl355_merkle_build(ctx.h, leaves.data_ptr(), n, L, cap_h, dig.data_ptr(), cap.data_ptr()))
    ctx.sync()
    cap_np = cap.cpu().numpy().view(np.uint64)
    dig_np = dig.cpu().numpy().view(np.uint64)
    leaves_np = leaves.cpu().numpy().view(np.uint64)
    want_dig, want_cap = orc.merkle_build(leaves_np, cap_h)
    eq(cap_np, want_cap)
    eq(dig_np, want_dig)
    rng = np.random.default_rng(1)
    sib = np.empty((18, 4), dtype=np.uint64)
    for i in np.concatenate([[0, 1, n - 1], rng.integers(0, n, 13)]):
        i = int(i)
        ctx.check(ctx.lib.gl355_merkle_prove(ctx.h, dig.data_ptr(), n, cap_h, i, sib.ctypes.data))
        assert orc.merkle_verify(leaves_np[i], i, sib, cap_np, cap_h), "path of leaf %d does not verify against the cap" % i
    del leaves, dig, cap, leaves_np, want_dig, dig_np
    torch.cuda.empty_cache()


# ---- cfg-2 (C) -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [21, 22, 23])
def test_ntt_above_2p20(ctx, orc, log_n):
    rng = np.random.default_rng(0x355 + log_n)
    x = rand_field(rng, (2, 1 << log_n))
    y = ctx.fft(x)
    eq(y, orc.ntt(x))
    eq(ctx.ifft(y), x)
    if log_n == 21:
        eq(ctx.coset_fft(x[:1]), orc.ntt(x[:1], shift=7))
        eq(ctx.coset_ifft(x[:1]), orc.ntt(x[:1], inverse=True, shift=7))


@pytest.mark.parametrize("log_n", [18, 20])
def test_lde_to_2p21_2p23(ctx, orc, log_n):
    rng = np.random.default_rng(0x455 + log_n)
    c = rand_field(rng, (2, 1 << log_n))
    want = orc.lde(c, 3)
    eq(ctx.lde(c, 3), want)
    br = ctx.lde(c, 3, bitrev=True)
    eq(ctx.reverse_index_bits(br[0]), want[0])


@pytest.mark.parametrize("log_n,rate_bits", [(15, 4), (16, 2), (18, 3), (19, 2), (19, 3), (15, 1)])
def test_lde_two_pass_coset_shapes(ctx, orc, log_n, rate_bits):
    """two-pass LDEs around the table-size limits of the all-cosets column pass: 16 / 4 / 2 cosets in one block, the largest sizes
    that still get full pre tables (8 x 2^18, 4 x 2^19), and the first one that falls back to the per-element power tables (8 x 2^19)"""
    rng = np.random.default_rng(0x456 + 8 * log_n + rate_bits)
    c = rand_field(rng, (2, 1 << log_n))
    c[0, :5] = [0, 0xFFFFFFFF00000000, 1, 0xFFFFFFFF00000000, 0]                      # zeros and p - 1 among the coefficients
    want = orc.lde(c, rate_bits)
    eq(ctx.lde(c, rate_bits), want)
    eq(ctx.lde(c, rate_bits, bitrev=True), orc.reverse_index_bits(want.T.copy()).T)


def test_lde_bench_shape_full_batch(gl, ctx, orc):
    """BASELINE configs[1] at its full size: 135 columns, 2^17 -> 2^20, bit-reversed output, operands resident (what bench.py times):
    EVERY column against the oracle's LDE (in slices of 27 columns to bound the host memory)"""
    import ctypes as C
    import torch
    rng = np.random.default_rng(0x457)
    B, log_n, rb = 135, 17, 3
    c = rand_field(rng, (B, 1 << log_n))
    cd = torch.from_numpy(c.view(np.int64)).cuda()
    out = torch.empty((B, 1 << (log_n + rb)), dtype=torch.int64, device="cuda")
    ctx.check(ctx.lib.gl355_lde_bitrev(ctx.h, C.c_void_p(cd.data_ptr()), log_n, rb, 7, B, C.c_void_p(out.data_ptr())))
    ctx.sync()
    got = out.cpu().numpy().view(np.uint64)
    for j0 in range(0, B, 27):
        want = orc.reverse_index_bits(orc.lde(c[j0:j0 + 27], rb).T.copy()).T
        for k in range(want.shape[0]):
            assert np.array_equal(got[j0 + k], want[k]), "column %d" % (j0 + k)


# ---- blinding stream -------------------------------------------------------------------------------------------------------
def test_blinding_stream_kernel(ctx, orc):
    for seed, stream, count in ((1, 1, 4 << 16), (0xDEADBEEF << 64, 4, 6803 * 135 + 41), (99, 3, 5), (7, 2, 1)):
        key = key_bytes(seed)
        got = np.empty(count, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_blinding_elements(ctx.h, key, stream, count, got.ctypes.data))
        want = np.empty(count, dtype=np.uint64)
        orc.L.orc_blinding_elements(key, C.c_uint32(stream), C.c_uint64(count), want.ctypes.data_as(C.c_void_p))
        eq(got, want)


def test_os_random_blinding_gives_distinct_valid_proofs(gl, ctx, orc):
    """blinding_key = NULL: the library draws a fresh key per proof (getrandom): two proofs of one witness differ and both verify"""
    import plonk_verifier as pv
    from test_gpu_prover import make_access_set
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x77)
    data, rows = aset.build(rng)
    topic = rand_field(rng, 4)
    idx, vals, pi = aset.witness_rows(rows, sks[2], topic, 2)
    a = plonk.prove_sparse(ctx, data, idx, vals, pi, None, flat_only=True)
    b = plonk.prove_sparse(ctx, data, idx, vals, pi, None, flat_only=True)
    assert not np.array_equal(a, b)
    for f in (a, b):
        proof = plonk.parse_proof(data, f)
        proof["public_inputs"] = pi
        pv.verify(orc, data.common(), proof)


# ---- cfg-4 at depth 20 -----------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def unit20(gl, ctx, orc):
    """the depth-20 unit on the CPU side (oracle) and the product's provers for the same access set"""
    import bench
    orc.L.orc_set_num_threads(bench.host_cores())
    cpu = cu.cpu_unit(orc)
    u = cu.UNIT_CASE
    pr = bench.RecursiveProvers(gl, 0, 2, log_members=u["log_members"], seed=u["seed"], replay_threads=2)
    return cpu, pr


def test_unit_depth20_byte_identical_and_golden(gl, ctx, orc, unit20):
    (case, topic, flat, pi, rc, outer, opis), pr = unit20
    u = cu.UNIT_CASE
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    assert np.array_equal(pr.sks, case["sks"]) and np.array_equal(pr.topic, topic) and np.array_equal(pr.root, case["root"])
    m = u["member"]
    g_in, g_pis = pr.sem.semaphore_prove(pr.sets[0], pr.sks[m], pr.topic, m, pr.aset.tree.prove_host(m), u["key_sem"])
    eq(g_pis, pi)
    eq(g_in, flat)
    g_out, g_opis = pr.nat.prove_tape(pr.sets[1], np.concatenate([g_in, g_pis]), u["key_rec"])
    eq(g_opis, opis)
    eq(g_out, outer)
    golden = json.load(open(os.path.join(HERE, "golden", "unit_depth20.json")))
    assert golden["case"] == u
    assert cs.digest_of(g_in) == golden["semaphore_sha256"] and cs.digest_of(g_out) == golden["recursive_sha256"]
    assert [int(x) for x in g_opis] == [int(x, 16) for x in golden["public_inputs"]]
    # the same unit through the native batch runtime (per-unit keys derived from a batch key): equal to the call-by-call proofs
    members = np.array([m, 0, (1 << 20) - 1], dtype=np.uint64)
    leaves, proofs, per = plonk.semaphore_units(pr.sets, pr.sem, pr.nat, pr.sks, pr.topic, pr.aset.tree.digests, members, 4242, want_proofs=True)
    for j, mm in enumerate(members):
        mm = int(mm)
        k0, k1 = plonk.derive_key(4242, 2 * j), plonk.derive_key(4242, 2 * j + 1)
        f, p = pr.sem.semaphore_prove(pr.sets[0], pr.sks[mm], pr.topic, mm, pr.aset.tree.prove_host(mm), k0)
        o, op = pr.nat.prove_tape(pr.sets[0], np.concatenate([f, p]), k1)
        eq(proofs[j], o)
        eq(leaves[j], op[4:12])
    # CPU restatement of unit j = 2 (the last member of the set: an all-ones index path)
    idx, vals, cpi = cs.witness(orc, case, (1 << 20) - 1, topic)
    c_in = case["cpu"].prove_sparse(idx, vals, cpi, plonk.derive_key(4242, 4))
    rows, ropis = cu.replay(rc, np.concatenate([c_in, cpi]))
    c_out = rc["cpu"].prove_sparse(rc["row_idx"], rows, ropis, plonk.derive_key(4242, 5))
    eq(proofs[2], c_out)


def test_fanin2_aggregation_node_byte_identical(gl, ctx, orc, unit20):
    """aggregate_signals (recursion.rs:25-185): a circuit verifying TWO signals (n = 2^15), GPU proof == CPU proof"""
    (case, topic, flat, pi, rc1, outer, opis), pr = unit20
    rec = importlib.import_module("stark-verifier_amd.recursion")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    sigs = []
    for j, m in enumerate((5, 77777)):
        f, p = pr.sem.semaphore_prove(pr.sets[0], pr.sks[m], pr.topic, m, pr.aset.tree.prove_host(m), 500 + j)
        sigs.append((f, np.asarray(p)))
    agg = rec.RecursiveCircuit(pr.sets[0], pr.inner_data.common(), k=2, public_inputs=rec.aggregate_public_inputs).build(sigs, np.random.default_rng(3))
    assert agg.data.degree_bits == 15
    g, gpis = agg.prove_flat(sigs, 600)
    crc = cu.recursive_cpu_circuit(orc, case["data"].common(), [s[0] for s in sigs], [s[1] for s in sigs], k=2)
    eq(crc["data"].circuit_digest, agg.data.circuit_digest)
    rows, cpis = cu.replay(crc, np.concatenate([np.concatenate([f, p]) for f, p in sigs]))
    eq(cpis, gpis)
    c = crc["cpu"].prove_sparse(crc["row_idx"], rows, cpis, 600)
    eq(g, c)
    # public inputs: root | nullifiers | topics (recursion.rs:105-165)
    eq(gpis[:4], case["root"])
    eq(gpis[4:12], np.concatenate([sigs[0][1][4:8], sigs[1][1][4:8]]))
