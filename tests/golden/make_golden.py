#!/usr/bin/env python3
"""Mints tests/golden/*.json.

  poseidon_kat.json   the upstream plonky2 Poseidon-Goldilocks known-answer vectors (SURVEY.md 8(c);
                      they are what plonky2's own poseidon_goldilocks tests assert), re-derived here
                      by the big-integer model from the reference's constants -- the script refuses
                      to write if the model disagrees with the transcribed expected words.
  conventions.json    tiny convention vectors + small NTT / LDE / Merkle / FRI
                      cases computed FROM THE DEFINITION by tests/pymodel.py (no oracle, no GPU).
  semaphore_proof.json  SHA-256 of the flat proof of one depth-2 Semaphore signal (tests/cpu_semaphore.py GOLDEN_CASE),
                      minted by the CPU restatement of prove() AFTER the restated reference verifier accepted it; the
                      CPU suite re-derives it, the GPU suite requires the product's proof to hash to the same value.
  unit_depth20.json   SHA-256 of the two proofs of one depth-20 unit (tests/cpu_unit.py UNIT_CASE), same rule.
  semaphore_depth25.json  the 2^25-member group of the reference's sweep (tests/cpu_semaphore.py GROUP25_CASE): root, public inputs, proof
                      SHA-256; `python tests/golden/make_golden.py depth25` (minutes: 67 M CPU permutations), not part of the default run.
Run: python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pymodel as pm  # noqa: E402

P = pm.P


def hx(v):
    return ["%016x" % x for x in v]


UPSTREAM = {
    "zeros": ([0] * 12, "3c18a9786cb0b359 c4055e3364a246c3 7953db0ab48808f4 c71603f33a1144ca d7709673896996dc 46a84e87642f44ed "
                        "d032648251ee0b3c 1c687363b207df62 df8565563e8045fe 40f5b37ff4254dae d070f637b431067c 1792b1c4342109d7"),
    "iota": (list(range(12)), "d64e1e3efc5b8e9e 53666633020aaa47 d40285597c6a8825 613a4f81e81231d2 414754bfebd051f0 cb1f8980294a023f "
                               "6eb2a9e4d54a9d0f 1902bc3af467e056 f045d5eafdc6021f e4150f77caaa3be5 c9bfd01d39b50cce 5c0a27fcb0e1459b"),
    "neg_one": ([P - 1] * 12, "be0085cfc57a8357 d95af71847d05c09 cf55a13d33c1c953 95803a74f4530e82 fcd99eb30a135df1 e095905e913a3029 "
                               "de0392461b42919b 7d3260e24e81d031 10d3d0465d9deaa0 a87571083dfc2a47 e18263681e9958f8 e28e96f1ae5e60d3"),
}


def mint_semaphore_proof():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import cpu_semaphore as cs
    import plonk_verifier as pv
    from oracle_lib import Oracle
    orc = Oracle()
    case, topic, (idx, vals, pi), flat = cs.golden_proof(orc)
    proof = case["plonk"].parse_proof(case["data"], flat)
    proof["public_inputs"] = pi
    pv.verify(orc, case["data"].common(), proof)          # refuse to mint a proof the restated reference verifier rejects
    json.dump({"case": cs.GOLDEN_CASE, "words": int(flat.size), "sha256": cs.digest_of(flat),
               "public_inputs": ["%016x" % int(x) for x in pi]}, open(os.path.join(HERE, "semaphore_proof.json"), "w"), indent=1)


def mint_unit_depth20():
    """unit_depth20.json: BASELINE configs[3] at its stated size -- one depth-20 Semaphore signal (2^20 members, signer 12) and the
    recursive proof verifying it, both from the CPU restatement of prove(), minted only after the restated reference verifier
    accepted both.  The GPU suite must reproduce both proofs byte for byte (tests/test_gpu_large.py)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import cpu_semaphore as cs
    import cpu_unit as cu
    import plonk_verifier as pv
    from oracle_lib import Oracle
    orc = Oracle()
    case, topic, flat, pi, rc, outer, opis = cu.cpu_unit(orc)
    for data, f, p in ((case["data"], flat, pi), (rc["data"], outer, opis)):
        proof = case["plonk"].parse_proof(data, f)
        proof["public_inputs"] = p
        pv.verify(orc, data.common(), proof)
    json.dump({"case": cu.UNIT_CASE, "semaphore_words": int(flat.size), "semaphore_sha256": cs.digest_of(flat),
               "recursive_words": int(outer.size), "recursive_sha256": cs.digest_of(outer), "recursive_degree_bits": int(rc["data"].degree_bits),
               "public_inputs": ["%016x" % int(x) for x in opis]}, open(os.path.join(HERE, "unit_depth20.json"), "w"), indent=1)


def mint_group_depth25():
    """semaphore_depth25.json: the LARGEST group of the reference's own size sweep (access_set.rs:193-215, `for pow in 20..26`): 2^25 members,
    signer 12 -- the group root, the signal's public inputs and the SHA-256 of the make_signal proof from the CPU restatement of prove(), minted
    only after the restated reference verifier accepted it.  ~5 minutes on 8 cores (67 M permutations for keys + tree); run with
    `python tests/golden/make_golden.py depth25`."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import cpu_semaphore as cs
    import plonk_verifier as pv
    from oracle_lib import Oracle, rand_field
    orc = Oracle()
    g = cs.GROUP25_CASE
    case = cs.build_case(orc, g["log_members"], g["seed"])
    topic = rand_field(case["rng"], 4)
    idx, vals, pi = cs.witness(orc, case, g["member"], topic)
    flat = case["cpu"].prove_sparse(idx, vals, pi, g["proof_seed"])
    proof = case["plonk"].parse_proof(case["data"], flat)
    proof["public_inputs"] = pi
    pv.verify(orc, case["data"].common(), proof)
    json.dump({"case": g, "words": int(flat.size), "sha256": cs.digest_of(flat), "root": ["%016x" % int(x) for x in case["root"]],
               "degree_bits": int(case["data"].degree_bits), "public_inputs": ["%016x" % int(x) for x in pi]},
              open(os.path.join(HERE, "semaphore_depth25.json"), "w"), indent=1)


def mint_bn254_kat():
    """poseidon_bn254_kat.json: (1) the published circomlib known answer poseidon([1,2,3,4]) for t = 5 -- the script refuses to
    write unless the big-integer model with the reference's parameters reproduces it; (2) permutation / hash vectors of the
    reference's Goldilocks-packed hasher (bn245_poseidon/plonky2_config.rs:38-75) from that pinned model."""
    import pymodel_bn254 as mb
    circomlib = "299c867db6c1fdd79dcefa40e4510b9837e60ebb1ce0663dbaa525df65250465"
    out = mb.permute_fr([0, 1, 2, 3, 4])
    assert "%064x" % out[0] == circomlib, "model + reference parameters do not reproduce the circomlib known answer"
    rnd = random.Random(0x254)
    kat = {"circomlib_poseidon_1_2_3_4": circomlib,
           "permute_fr": [{"input": ["%064x" % v for v in [0, 1, 2, 3, 4]], "output": ["%064x" % v for v in out]}],
           "permute": [], "hash_no_pad": [], "two_to_one": []}
    for name, st in (("zeros", [0] * 12), ("iota", list(range(12))), ("neg_one", [P - 1] * 12), ("random", [rnd.randrange(P) for _ in range(12)])):
        kat["permute"].append({"name": name, "input": hx(st), "output": hx(mb.permute(st))})
    for k in (1, 4, 7, 8, 9, 16, 135):
        xs = [rnd.randrange(P) for _ in range(k)]
        kat["hash_no_pad"].append({"input": hx(xs), "output": hx(mb.hash_no_pad(xs))})
    l, r = [rnd.randrange(P) for _ in range(4)], [rnd.randrange(P) for _ in range(4)]
    kat["two_to_one"].append({"left": hx(l), "right": hx(r), "output": hx(mb.two_to_one(l, r))})
    json.dump(kat, open(os.path.join(HERE, "poseidon_bn254_kat.json"), "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "depth25":
        return mint_group_depth25()
    mint_semaphore_proof()
    mint_unit_depth20()
    mint_bn254_kat()
    kat = {"permute": []}
    for name, (inp, want) in UPSTREAM.items():
        got = pm.permute(inp)
        assert hx(got) == want.split(), "model disagrees with upstream KAT %s" % name
        kat["permute"].append({"name": name, "input": hx(inp), "output": hx(got)})
    kat["hash_no_pad"] = [{"input": hx(list(range(1, k + 1))), "output": hx(pm.hash_no_pad(list(range(1, k + 1))))}
                          for k in (1, 7, 8, 9, 16, 135)]
    assert kat["hash_no_pad"][2]["output"] == "d110aa6a46373941 8f238fcceb658894 9cd4f8353866fb4f 274913f0007aa232".split()
    json.dump(kat, open(os.path.join(HERE, "poseidon_kat.json"), "w"), indent=1)

    rnd = random.Random(0x355)
    conv = {"omega_8": "%016x" % pm.root_of_unity(3), "omega_2_16": "%016x" % pm.root_of_unity(16),
            "omega_2_32": "%016x" % pm.root_of_unity(32)}
    assert conv["omega_8"] == "fffffffeff000001" and conv["omega_2_32"] == "185629dcda58878c"
    conv["ntt"] = []
    for lg in (0, 1, 3, 5, 7):
        c = list(range(1, 9)) if lg == 3 else [rnd.randrange(P) for _ in range(1 << lg)]
        conv["ntt"].append({"input": hx(c), "forward": hx(pm.dft(c)), "inverse": hx(pm.dft(c, inverse=True))})
    conv["lde"] = []
    for lg, rb in ((2, 1), (3, 3), (5, 3), (4, 2)):
        c = [1, 2, 3, 4] if lg == 2 else [rnd.randrange(P) for _ in range(1 << lg)]
        conv["lde"].append({"coeffs": hx(c), "rate_bits": rb, "shift": 7, "natural": hx(pm.lde(c, rb))})
    conv["merkle"] = []
    for n, ll, cap in ((8, 4, 0), (8, 4, 1), (8, 9, 0), (16, 3, 2), (4, 135, 1), (2, 4, 1)):
        leaves = [[4 * i + j for j in range(4)] for i in range(8)] if (n, ll) == (8, 4) else \
                 [[9 * i + j for j in range(9)] for i in range(8)] if (n, ll) == (8, 9) else \
                 [[rnd.randrange(P) for _ in range(ll)] for _ in range(n)]
        dig, capv = pm.merkle(leaves, cap)
        conv["merkle"].append({"leaves": [hx(l) for l in leaves], "cap_height": cap,
                               "digests": [hx(d) for d in dig], "cap": [hx(c) for c in capv]})
    a, b = (3, 5), (P - 2, 11)
    conv["ext_mul"] = {"a": hx(a), "b": hx(b), "out": hx(pm.ext_mul(a, b))}
    json.dump(conv, open(os.path.join(HERE, "conventions.json"), "w"), indent=1)
    print("golden vectors written")


if __name__ == "__main__":
    main()
