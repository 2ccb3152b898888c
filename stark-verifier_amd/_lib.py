"""ctypes binding of libgl355.so (include/gl355.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgl355.so")

u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class Gl355Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gl355 error %d: %s" % (code, msg))
        self.code = code


class PolyRef(C.Structure):
    _fields_ = [("oracle", vp), ("column", C.c_uint32)]


class Challenger(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("in_buf", C.c_uint64 * 8), ("in_len", C.c_uint32),
                ("out_buf", C.c_uint64 * 8), ("out_len", C.c_uint32), ("hasher", C.c_int32)]


class VerifierData(C.Structure):
    _fields_ = [("circuit", vp), ("constants_sigmas_cap", vp), ("k_is", vp), ("circuit_digest", C.c_uint64 * 4),
                ("cap_height", C.c_uint32), ("pow_bits", C.c_uint32), ("num_queries", C.c_uint32), ("n_fri_layers", C.c_uint32),
                ("zero_knowledge", C.c_int32), ("hasher", C.c_int32)]


class ProverData(C.Structure):
    _fields_ = [("circuit", vp), ("constants_sigmas", vp), ("sigmas", vp), ("k_is", vp), ("circuit_digest", C.c_uint64 * 4),
                ("cap_height", C.c_uint32), ("pow_bits", C.c_uint32), ("num_queries", C.c_uint32), ("n_fri_layers", C.c_uint32),
                ("zero_knowledge", C.c_int32), ("hasher", C.c_int32)]


MAX_GATES = 16
(GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_BASE_SUM, GATE_POSEIDON, GATE_ARITHMETIC, GATE_ARITHMETIC_EXT,
 GATE_MUL_EXT, GATE_POSEIDON_MDS, GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT) = range(12)


class Gate(C.Structure):
    _fields_ = [("type", C.c_uint32), ("param", C.c_uint32), ("selector_index", C.c_uint32),
                ("group_start", C.c_uint32), ("group_end", C.c_uint32)]


class Circuit(C.Structure):
    _fields_ = [("degree_bits", C.c_uint32), ("rate_bits", C.c_uint32), ("num_wires", C.c_uint32),
                ("num_routed_wires", C.c_uint32), ("num_constants", C.c_uint32), ("num_selectors", C.c_uint32),
                ("num_challenges", C.c_uint32), ("max_degree", C.c_uint32), ("num_partial_products", C.c_uint32),
                ("num_gates", C.c_uint32), ("gates", Gate * MAX_GATES)]


# name -> (restype, argtypes).  Data pointers are void* so numpy arrays, torch data_ptr() ints and
# raw device pointers all pass through the same signature.
SIGNATURES = {
    "gl355_runtime_config": (C.c_int32, [C.c_int32, C.c_uint32, C.c_int32]),
    "gl355_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(vp)]),
    "gl355_ctx_create_on_stream": (C.c_int32, [C.c_int32, vp, C.POINTER(vp)]),
    "gl355_ctx_destroy": (C.c_int32, [vp]),
    "gl355_ctx_sync": (C.c_int32, [vp]),
    "gl355_ctx_set_option": (C.c_int32, [vp, C.c_int32, C.c_int64]),
    "gl355_last_error": (C.c_char_p, [vp]),
    "gl355_version": (C.c_char_p, []),
    "gl355_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "gl355_aggregate_units": (C.c_int32, [vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint32, vp]),
    "gl355_keccak256": (C.c_int32, [vp, C.c_uint64, vp]),
    "gl355_kzg_commit_columns": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "gl355_plonk_keygen": (C.c_int32, [vp, vp, C.c_uint64, vp, vp, vp, vp, C.POINTER(vp)]),
    "gl355_plonk_pk_info": (C.c_int32, [vp, vp]),
    "gl355_plonk_pk_commitments": (C.c_int32, [vp, vp, vp]),
    "gl355_plonk_pk_digest": (C.c_int32, [vp, vp]),
    "gl355_plonk_pk_set_digest": (C.c_int32, [vp, vp]),
    "gl355_plonk_pk_export_quotient": (C.c_int32, [vp, vp]),
    "gl355_plonk_prove": (C.c_int32, [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), vp, vp]),
    "gl355_plonk_pk_destroy": (C.c_int32, [vp]),
    "gl355_valu_probe": (C.c_int32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gl355_clock_probe": (C.c_int32, [vp, C.c_uint32, C.POINTER(C.c_double)]),
    "gl355_valu_probe_op_name": (C.c_char_p, [C.c_uint32]),
    "gl355_valu_probe_composite_name": (C.c_char_p, [C.c_uint32]),
    "gl355_valu_probe_pair_names": (C.c_int32, [C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "gl355_valu_probe_pairs": (C.c_int32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gl355_valu_probe_ops": (C.c_int32, [vp, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gl355_valu_probe_composite": (C.c_int32, [vp, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "gl355_malloc": (C.c_int32, [vp, C.c_size_t, C.POINTER(vp)]),
    "gl355_free": (C.c_int32, [vp, vp]),
    "gl355_memcpy_h2d": (C.c_int32, [vp, vp, vp, C.c_size_t]),
    "gl355_memcpy_d2h": (C.c_int32, [vp, vp, vp, C.c_size_t]),
    "gl355_timer_start": (C.c_int32, [vp]),
    "gl355_timer_stop": (C.c_int32, [vp, C.POINTER(C.c_float)]),
    "gl355_profile_enable": (C.c_int32, [vp, C.c_int32]),
    "gl355_profile_read": (C.c_int32, [vp, C.c_char_p, C.c_size_t]),
    "gl355_field_batch": (C.c_int32, [vp, C.c_int32, vp, vp, vp, C.c_uint64]),
    "gl355_ntt": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int32]),
    "gl355_coset_ntt": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32]),
    "gl355_lde": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, vp]),
    "gl355_lde_bitrev": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, vp]),
    "gl355_transpose": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, vp]),
    "gl355_reverse_index_bits": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32]),
    "gl355_poseidon_permute": (C.c_int32, [vp, vp, C.c_uint64]),
    "gl355_hash_no_pad": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp]),
    "gl355_hash_leaves": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp]),
    "gl355_two_to_one": (C.c_int32, [vp, vp, vp, C.c_uint64, vp]),
    "gl355_merkle_build": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp]),
    "gl355_merkle_prove": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint64, vp]),
    "gl355_commit_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp, C.c_uint32, C.POINTER(vp)]),
    "gl355_commit": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp, C.c_uint32, C.POINTER(vp)]),
    "gl355_oracle_destroy": (C.c_int32, [vp]),
    "gl355_oracle_info": (C.c_int32, [vp] + [C.POINTER(C.c_uint32)] * 5),
    "gl355_oracle_cap": (C.c_int32, [vp, vp]),
    "gl355_oracle_coeffs": (C.c_int32, [vp, vp]),
    "gl355_oracle_leaves": (C.c_int32, [vp, vp]),
    "gl355_oracle_digests": (C.c_int32, [vp, vp]),
    "gl355_oracle_lde_ptr": (vp, [vp]),
    "gl355_oracle_coeffs_ptr": (vp, [vp]),
    "gl355_oracle_open": (C.c_int32, [vp, C.c_uint64, vp, vp]),
    "gl355_oracle_open_batch": (C.c_int32, [vp, vp, C.c_uint32, vp, vp]),
    "gl355_fri_prove": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(Challenger), vp, vp, C.POINTER(C.c_uint64), vp, vp, vp]),
    "gl355_proof_words": (C.c_uint64, [C.POINTER(ProverData)]),
    "gl355_prove": (C.c_int32, [vp, C.POINTER(ProverData), vp, vp, C.c_uint32, vp, vp, C.c_uint64]),
    "gl355_derive_key": (C.c_int32, [C.c_char_p, C.c_uint64, C.c_char_p]),
    "gl355_blinding_elements": (C.c_int32, [vp, C.c_char_p, C.c_uint32, C.c_uint64, vp]),
    "gl355_prove_sparse": (C.c_int32, [vp, C.POINTER(ProverData), vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       vp, C.c_uint32, vp, vp, C.c_uint64]),
    "gl355_prove_sparse_units": (C.c_int32, [vp, C.POINTER(ProverData), C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             vp, C.c_uint32, vp, vp, C.c_uint64]),
    "gl355_permute_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint64]),
    "gl355_hash_no_pad_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint64, C.c_uint32, vp]),
    "gl355_hash_leaves_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint64, C.c_uint32, vp]),
    "gl355_two_to_one_h": (C.c_int32, [vp, C.c_int32, vp, vp, C.c_uint64, vp]),
    "gl355_merkle_build_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp]),
    "gl355_circuit_load": (C.c_int32, [vp, vp, C.c_uint64, C.POINTER(vp)]),
    "gl355_circuit_destroy": (C.c_int32, [vp]),
    "gl355_circuit_info": (C.c_int32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint32)]),
    "gl355_circuit_digest": (C.POINTER(C.c_uint64), [vp]),
    "gl355_verify": (C.c_int32, [C.POINTER(VerifierData), vp, C.c_uint64, vp, C.c_uint32]),
    "gl355_verify_last_error": (C.c_char_p, []),
    "gl355_circuit_verify": (C.c_int32, [vp, vp, C.c_uint64, vp, C.c_uint32]),
    "gl355_circuit_witness_rows": (C.c_int32, [vp, vp, C.c_uint32, vp, C.c_uint64, C.c_int32, vp, vp, C.POINTER(C.c_uint64)]),
    "gl355_circuit_prove_rows": (C.c_int32, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64]),
    "gl355_circuit_prove_tape": (C.c_int32, [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp]),
    "gl355_semaphore_prove": (C.c_int32, [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint32, vp, vp, C.c_uint64, vp]),
    "gl355_circuit_prove_rows_units": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp]),
    "gl355_circuit_prove_tape_units": (C.c_int32, [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp]),
    "gl355_semaphore_prove_units": (C.c_int32, [vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, vp, vp, vp]),
    "gl355_semaphore_units": (C.c_int32, [vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint32, vp, vp, vp, vp]),
    "gl355_semaphore_witness": (C.c_int32, [vp, vp, C.c_uint64, vp, C.c_uint32, vp, vp]),
    "gl355_witness_replay_segmented": (C.c_int32, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, C.c_uint32,
                                                   C.POINTER(C.c_uint64)]),
    "gl355_witness_replay": (C.c_int32, [vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    "gl355_quotient": (C.c_int32, [vp, C.POINTER(Circuit), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gl355_quotient_values": (C.c_int32, [vp, C.POINTER(Circuit), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gl355_challenger_init": (C.c_int32, [C.POINTER(Challenger)]),
    "gl355_challenger_init_h": (C.c_int32, [C.POINTER(Challenger), C.c_int32]),
    "gl355_challenger_observe": (C.c_int32, [C.POINTER(Challenger), vp, C.c_uint64]),
    "gl355_challenger_squeeze": (C.c_int32, [C.POINTER(Challenger), vp, C.c_uint64]),
    "gl355_challenger_pow_state": (C.c_int32, [C.POINTER(Challenger), vp, C.POINTER(C.c_uint32)]),
    "gl355_host_poseidon_permute": (C.c_int32, [vp]),
    "gl355_host_hash_no_pad": (C.c_int32, [vp, C.c_uint64, vp]),
    "gl355_host_hash_no_pad_h": (C.c_int32, [C.c_int32, vp, C.c_uint64, vp]),
    "gl355_host_permute_h": (C.c_int32, [C.c_int32, vp]),
    "gl355_poseidon_gate_witness": (C.c_int32, [vp, C.c_uint64, vp]),
    "gl355_deep_batch": (C.c_int32, [vp, C.POINTER(PolyRef), C.c_uint32, vp, vp, vp]),
    "gl355_eval_polys": (C.c_int32, [vp, C.POINTER(PolyRef), C.c_uint32, vp, vp]),
    "gl355_lde_ext": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, vp]),
    "gl355_fri_fold": (C.c_int32, [vp, vp, C.c_uint64, vp, vp]),
    "gl355_fri_layer_commit_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "gl355_fri_layer_commit": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "gl355_pow_grind_h": (C.c_int32, [vp, C.c_int32, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "gl355_pow_grind": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "gl355_comm_unique_id": (C.c_int32, [C.c_int32, C.c_char_p]),
    "gl355_comm_host_id": (C.c_int32, [C.c_char_p, C.c_uint16, C.c_char_p]),
    "gl355_comm_create": (C.c_int32, [vp, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "gl355_comm_destroy": (C.c_int32, [vp]),
    "gl355_comm_info": (C.c_int32, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gl355_comm_last_error": (C.c_char_p, [vp]),
    "gl355_gather_digests": (C.c_int32, [vp, vp, C.c_uint64, vp]),
    "gl355_comm_barrier": (C.c_int32, [vp]),
    "gl355_comm_max_f64": (C.c_int32, [vp, C.POINTER(C.c_double)]),
    "gl355_aggregation_root": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp]),
    "gl355_bn254_fr_ntt": (C.c_int32, [vp, vp, C.c_uint32, C.c_int32]),
    "gl355_bn254_fr_coset_ntt": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_int32, vp]),
    "gl355_bn254_g1_msm": (C.c_int32, [vp, vp, vp, C.c_uint64, vp]),
    "gl355_bn254_g1_msm_batch": (C.c_int32, [vp, vp, vp, C.c_uint64, C.c_uint32, vp]),
    "gl355_bn254_g1_msm_prepare": (C.c_int32, [vp, vp, C.c_uint64, C.POINTER(vp)]),
    "gl355_bn254_g1_msm_prepared": (C.c_int32, [vp, vp, vp, C.c_uint32, vp]),
    "gl355_bn254_g1_msm_bases_free": (C.c_int32, [vp, vp]),
    "gl355_bn254_g1_fixed_base_mul": (C.c_int32, [vp, vp, vp, C.c_uint64, vp]),
    "gl355_kzg_setup": (C.c_int32, [vp, vp, C.c_uint32, vp, vp]),
    "gl355_kzg_commit": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_int32, vp]),
    "gl355_kzg_open": (C.c_int32, [vp, vp, vp, C.c_uint32, vp, vp, vp, vp]),
    "gl355_zs_partial_products": (C.c_int32, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                              C.c_uint64, vp, vp]),
}

_lib = None


def load(init_torch=True):
    """Load libgl355.so and attach signatures.  Raises if the library has not been built.
    init_torch=False: do not initialise torch's CUDA state first -- for callers that must run gl355_runtime_config before the
    HIP runtime starts (torch has to be imported already, so that its libamdhip64 is the one in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgl355.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C stark-verifier_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # If torch is in this process it must own the HIP runtime initialisation (it ships its own ROCm
    # libraries; loading the system libamdhip64 first makes torch report "No HIP GPUs are available").
    import sys
    if init_torch and "torch" in sys.modules:
        try:
            t = sys.modules["torch"]
            t.cuda.is_available() and t.cuda.init()
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
