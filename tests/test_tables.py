"""CPU: the derived Poseidon fast-partial-round tables (tools/gen_poseidon_tables.py)."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon_tables as gpt  # noqa: E402


def test_fast_equals_naive_and_header_is_current():
    rc = gpt.load_rc()
    tb = gpt.derive(rc)
    rnd = random.Random(11)
    for _ in range(5):
        st = [rnd.randrange(gpt.P) for _ in range(12)]
        assert gpt.permute_fast(st, rc, tb) == gpt.permute_naive(st, rc)
    assert tb["post"][21] == 0 and tb["first"][0] == rc[4][0]
    hdr = open(os.path.join(ROOT, "stark-verifier_amd", "csrc", "poseidon_tables.h")).read()
    for v in (tb["first"][5], tb["post"][7], tb["vs"][3][4], tb["w_hats"][20][10], tb["init"][10][10]):
        assert "0x%016x" % v in hdr


def test_tables_equal_reference_literals():
    if not os.path.exists("/root/reference/src/plonky2_verifier/chip/plonk/gates/poseidon.rs"):
        pytest.skip("reference tree not present on this box")
    assert gpt.check_reference(gpt.derive(gpt.load_rc()))


def test_block_form_equals_naive_and_header_is_current():
    """The block form of the 22 partial rounds (poseidon.cuh, psd_partial_rounds_block): the limb-exact model of the device algorithm (22-bit limbs,
    two 64-bit accumulators per dot product, accumulator bounds asserted) against the naive permutation -- edge states, random states, and the
    upstream known answer -- and the committed header holds exactly the tables the generator derives."""
    import re
    rc = gpt.load_rc()
    kt = gpt.kform_tables(rc)
    assert kt["B"] == 11 and len(kt["mac"]) == 429 and len(kt["add"]) == 2 * 22
    rnd = random.Random(12)
    P = gpt.P
    states = [[0] * 12, [P - 1] * 12, [(1 << 32) - 1] * 12, list(range(12))] + [[rnd.randrange(P) for _ in range(12)] for _ in range(6)]
    for st in states:
        want = gpt.permute_naive(st, rc)
        assert gpt.permute_kform(st, rc, kt) == want
        assert gpt.permute_kform(st, rc, kt, noncanonical=True) == want
    assert gpt.permute_kform([0] * 12, rc, kt)[0] == 0x3c18a9786cb0b359            # plonky2's test vector for the all-zero state
    hdr = open(os.path.join(ROOT, "stark-verifier_amd", "csrc", "poseidon_ktables.h")).read()
    nums = [int(x, 16) for x in re.findall(r"UINT64_C\((0x[0-9a-f]+)\)", hdr)]
    want = [v for c in kt["mac"] for v in gpt.k_triple(c)] + [v for c in kt["add"] for v in (c & 0xFFFFFFFF, c >> 32)]
    assert nums == want
    # a multiply-accumulate's three words are c, 2^22 c, 2^44 c mod p, and the worst-case accumulator (66 products + a 32-bit constant) fits 64 bits
    assert all(nums[3 * i + 1] == (nums[3 * i] << 22) % P and nums[3 * i + 2] == (nums[3 * i] << 44) % P for i in range(429))
    assert 66 * ((1 << 22) - 1) * ((1 << 32) - 1) + (1 << 32) < (1 << 61)
