"""gl355 -- MI355X-native prover hot path behind the reference's plonky2 call sites.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it with
    import importlib; gl = importlib.import_module("stark-verifier_amd")
`api` mirrors the reference-side interface (plonky2 names); `_lib` is the raw ctypes binding of
include/gl355.h.  The HIP library is required: nothing in this package computes on the CPU.
"""
from . import _lib, api  # noqa: F401
from .api import (COSET_SHIFT, HASH_BN254_POSEIDON, HASH_POSEIDON, P, SALT_SIZE, Bn254PoseidonHash, Context, MerkleTree,  # noqa: F401
                  PolynomialBatch, PoseidonHash,
                  deep_batch, eval_polys, rand_field)
from ._lib import Gl355Error  # noqa: F401
