/*
 * gl355.h -- C ABI of libgl355, the MI355X (gfx950) prover hot path behind the reference's
 * plonky2 call sites.
 *
 * The reference (DoHoonKim8/stark-verifier) has no FFI of its own: the seam is the Cargo edge into
 * plonky2 @ 72229c47 (Cargo.lock:1573-1612).  Each entry point below names the plonky2 item it
 * replaces and the reference call site that reaches it; INTEGRATION.md shows the Rust `extern "C"`
 * shim a maintainer would patch into plonky2 to bind them.
 *
 * Conventions
 *   - Field elements are uint64_t, little-endian; ANY u64 is accepted on input (reduced mod
 *     p = 2^64 - 2^32 + 1), outputs are canonical (< p).  Extension elements (F_p[X]/(X^2-7)) are
 *     two consecutive u64 (c0, c1).
 *   - Every data pointer may be a HOST pointer or a DEVICE pointer (hipMalloc / gl355_malloc / a
 *     torch tensor's data_ptr).  Host buffers are staged through HBM by the library (the call
 *     returns after the result is back in the host buffer); device buffers are used in place and
 *     the call returns after enqueueing on the context's stream (use gl355_ctx_sync).
 *   - The caller owns every buffer; the library never frees caller memory.
 *   - Every function returns int32_t: 0 = OK, < 0 = GL355_E_*.  No exception or abort crosses
 *     the boundary.  gl355_last_error(ctx) gives the text of the last failure on that context.
 *   - A gl355_ctx is bound to one device and one stream; contexts are independent, so the API is
 *     re-entrant from many host threads (the reference calls prove() from rayon workers,
 *     src/plonky2_semaphore/recursion.rs:214-227,300-308) as long as each thread uses its own ctx.
 */
#ifndef GL355_H
#define GL355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL355_OK 0
#define GL355_E_INVALID_ARG (-1)
#define GL355_E_NO_DEVICE (-2)
#define GL355_E_OOM (-3)
#define GL355_E_HIP (-4)
#define GL355_E_UNSUPPORTED (-5)
#define GL355_E_WITNESS (-6)     /* witness generation hit an unsatisfiable constraint (invalid inner proof) */
#define GL355_E_VERIFY (-7)      /* gl355_verify: the proof is not valid for this circuit / these public inputs */

#define GL355_P UINT64_C(0xFFFFFFFF00000001) /* chip/native_chip/arithmetic_chip.rs:19 */
#define GL355_COSET_SHIFT UINT64_C(7)        /* chip/plonk/plonk_verifier_chip.rs:225-227 */
#define GL355_SALT_SIZE 4                    /* types/assigned.rs:67-71 */
#define GL355_MAX_UNITS 16                   /* independent proofs of one circuit a prover context proves in lock-step (gl355_*_units) */

typedef struct gl355_ctx gl355_ctx;
typedef struct gl355_oracle gl355_oracle; /* a committed polynomial batch resident in HBM */

/* ---- context ------------------------------------------------------------------------------ */
/* Process-level runtime settings for the many-provers-per-GPU regime of the reference's rayon loop (recursion.rs:214-227,
 * 300-308); call ONCE PER DEVICE BEFORE anything initialises the HIP runtime for it (first gl355_ctx_create, torch, ...):
 *   contexts > 0      two hardware queues per prover context -- its proving stream and the batch runtime's side stream
 *                     (GPU_MAX_HW_QUEUES = 2 * contexts, unless the variable is already set);
 *   sleeping_waits    hipDeviceScheduleBlockingSync: every device wait of the process sleeps on the completion interrupt
 *                     instead of spinning -- needed when there are more contexts than usable cores.
 * GL355_E_HIP if the runtime refuses (typically: the device was already initialised). */
int32_t gl355_runtime_config(int32_t device, uint32_t contexts, int32_t sleeping_waits);
int32_t gl355_ctx_create(int32_t device, gl355_ctx** out);
/* use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int32_t gl355_ctx_create_on_stream(int32_t device, void* hip_stream, gl355_ctx** out);
int32_t gl355_ctx_destroy(gl355_ctx* ctx);
int32_t gl355_ctx_sync(gl355_ctx* ctx);
/* tuning knobs (results never depend on them).  MERKLE_LANES_LOG: Merkle levels with at most 2^value nodes run the
 * 16-lanes-per-node kernel (lowest latency, ~3x the instructions of the one-lane-per-node kernel); default 14 suits a
 * single proof stream, 11..12 gives more proofs/s when many contexts share the GPU. */
enum { GL355_OPT_MERKLE_LANES_LOG = 1,
       GL355_OPT_BLOCKING_SYNC = 2,     /* how the context's host thread waits for its stream: 0 hipStreamSynchronize (spins unless the
                                           device runs gl355_runtime_config(.., sleeping_waits)), 1 a blocking event, 2 poll + back-off
                                           (hipStreamQuery, 30-us sleeps): a few percent of a core per waiting context and ~30 us of
                                           wake-up latency -- for ranks with more prover contexts than cores; 3 poll without sleeping (a lone proof's
                                           latency setting: its ~40 waits each end within one poll of the completion) */
       GL355_OPT_REPLAY_THREADS = 3,    /* host threads gl355_circuit_prove_tape uses for a segmented tape (default 1) */
       GL355_OPT_NTT_SINGLE_PASS_MAX_LOG = 4, /* 12..14 (default 12): commit-path transforms of 2^13 / 2^14 points above this size run
                                           in two passes (a streaming 2- / 4-row column pass, then 4096-point limb rows, 3 tiles per
                                           CU) instead of one pass whose 64- / 128-KB tile owns a whole CU */
       GL355_OPT_BATCH_UNITS = 5,       /* 1..GL355_MAX_UNITS (default 8): units gl355_semaphore_units proves in lock-step per context */
       GL355_OPT_DEVICE_REPLAY = 6 };   /* != 0 (default): gl355_semaphore_units generates the recursive circuit's witness rows on the
                                           device (tape interpreter kernel) instead of on host threads; same rows, same failures */
int32_t gl355_ctx_set_option(gl355_ctx* ctx, int32_t option, int64_t value);
const char* gl355_last_error(gl355_ctx* ctx);
const char* gl355_version(void);
int32_t gl355_device_count(int32_t* out);

/* ---- the VALU roofline's peak, measured on the device in the run that reports it (csrc/valu_probe.hip).  Poseidon / Merkle /
 * the constraint kernel are integer-VALU bound ("Roofline: integer VALU issue ... not HBM and not MFMA"); their ceiling is
 * the chip's issue rate for their instruction mix.
 *   gl355_valu_probe   per class one kernel that only issues that instruction (8 independent chains per lane, 8 waves per SIMD, all SIMDs):
 *                      rates[c] = wave-level instructions per second in units of 1e9, shader_mhz[c] = the shader clock read inside that
 *                      kernel.  A kernel whose instructions split as f_c has the peak 1 / sum_c (f_c / rates[c]).
 *   gl355_clock_probe  shader clock (MHz) over `micros` microseconds of one sleeping wave on the context's stream: the clock under whatever
 *                      else the device runs meanwhile */
enum { GL355_VALU_FULL32 = 0 /* v_add_u32: add / sub / logic / right shift / move */, GL355_VALU_HALF32 = 1 /* v_add_co_u32: carries, left shifts,
       v_mul_lo, three-operand forms */, GL355_VALU_MAD64 = 2 /* v_mad_u64_u32 and the 64-bit shifts */, GL355_VALU_CLASSES = 3 };
int32_t gl355_valu_probe(gl355_ctx* ctx, double rates_ginst_per_s[GL355_VALU_CLASSES], double shader_mhz[GL355_VALU_CLASSES]);
int32_t gl355_clock_probe(gl355_ctx* ctx, uint32_t micros, double* shader_mhz);
/* Round 6: the ceiling priced per OPCODE FORM (the three classes above proved too coarse: the job issued faster than their harmonic combination).
 *   gl355_valu_probe_ops        one kernel per opcode form the library ships (gl355_valu_probe_op_name(i), i < GL355_VALU_PROBE_OPS; NULL beyond), each
 *                               with `ilp` = 1, 4 or 8 independent dependency chains per lane at 8 waves per SIMD on every SIMD: rates[i] in 1e9 wave
 *                               instructions per second, shader_mhz[i] read inside that kernel.  cost_i = 1024 SIMDs x MHz / rate, in shader cycles
 *                               per wave instruction per SIMD; a kernel whose instructions split as f_i cannot issue faster than
 *                               1024 x clock / sum_i f_i cost_i (tools/isa_mix.py has the f_i of every shipped kernel).
 *   gl355_valu_probe_composite  the shipped code on register operands, no memory: 0 = the field product (four in lock-step, gl_mul_multi<4>), items =
 *                               products; 1 = the Poseidon permutation at the hash kernels' occupancy, items = permutations; 2 = v_mad_u64_u32 and
 *                               v_add_u32 alternating in one wave, items = pairs; 3 = the same two on different waves of a SIMD, items = instructions;
 *                               4 .. 12 = runs of two opcode forms alternating in one wave (gl355_valu_probe_composite_name(i)), items = instructions.
 *                               items_g_per_s in 1e9 lane-level items per second; waves_per_simd = the resident waves the probe ran with */
enum { GL355_VALU_PROBE_OPS = 25, GL355_VALU_PROBE_COMPOSITES = 13 };
const char* gl355_valu_probe_op_name(uint32_t i);
const char* gl355_valu_probe_composite_name(uint32_t i);
int32_t gl355_valu_probe_ops(gl355_ctx* ctx, uint32_t ilp, double rates_ginst_per_s[GL355_VALU_PROBE_OPS], double shader_mhz[GL355_VALU_PROBE_OPS]);
int32_t gl355_valu_probe_composite(gl355_ctx* ctx, uint32_t which, double* items_g_per_s, double* shader_mhz, uint32_t* waves_per_simd);
/*   gl355_valu_probe_pairs      one kernel per unordered pair (X, Y) of the twelve opcode forms that carry the job's instruction count
 *                               (gl355_valu_probe_pair_names(i, &x, &y), i < GL355_VALU_PROBE_PAIRS): runs of four X and four Y alternating in one wave,
 *                               8 waves per SIMD; rates[i] in 1e9 wave instructions per second (X and Y together).  2 x 1024 x MHz / rate = the cycles
 *                               one X and one Y take TOGETHER; where that is less than the sum of their stand-alone costs the two overlap, and a
 *                               ceiling has to price them so (bench.py: the cheapest pairing of a kernel's instructions, a small linear program) */
enum { GL355_VALU_PROBE_PAIRS = 66 };
int32_t gl355_valu_probe_pair_names(uint32_t i, const char** form_a, const char** form_b);
int32_t gl355_valu_probe_pairs(gl355_ctx* ctx, double rates_ginst_per_s[GL355_VALU_PROBE_PAIRS], double shader_mhz[GL355_VALU_PROBE_PAIRS]);

/* device memory helpers so a non-torch host (the Rust shim) can keep operands resident */
int32_t gl355_malloc(gl355_ctx* ctx, size_t bytes, void** dptr);
int32_t gl355_free(gl355_ctx* ctx, void* dptr);
int32_t gl355_memcpy_h2d(gl355_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int32_t gl355_memcpy_d2h(gl355_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* HIP-event timer on the context's stream (what bench.py times kernels with) */
int32_t gl355_timer_start(gl355_ctx* ctx);
int32_t gl355_timer_stop(gl355_ctx* ctx, float* ms);
/* per-kernel-group HIP-event timing: enable, run, then read "name count total_ms algorithmic_bytes" lines
 * (algorithmic bytes = the compulsory HBM reads + writes of those launches, the numerator of the roofline) */
int32_t gl355_profile_enable(gl355_ctx* ctx, int32_t on);
int32_t gl355_profile_read(gl355_ctx* ctx, char* buf, size_t buf_len);

/* ---- a1: GoldilocksField / QuadraticExtension (plonky2_field; signal.rs:1,5) ---------------- */
enum { GL355_OP_ADD = 0, GL355_OP_SUB = 1, GL355_OP_MUL = 2, GL355_OP_INV = 3,
       GL355_OP_EXT_MUL = 4, GL355_OP_EXT_INV = 5 };
/* out[i] = a[i] (op) b[i]; for EXT ops n counts extension elements; b ignored for *_INV */
int32_t gl355_field_batch(gl355_ctx* ctx, int32_t op, const uint64_t* a, const uint64_t* b,
                          uint64_t* out, uint64_t n);

/* ---- a2: fft_with_options / ifft_with_options (plonky2_field::fft) --------------------------
 * In place, natural order in and out.  batch columns, column c at data + c*stride, n = 2^log_n.
 * forward: out[j] = sum_i in[i] * omega_n^(i j), omega_n = 7^((p-1)/n)  (fri_chip.rs:162-163). */
int32_t gl355_ntt(gl355_ctx* ctx, uint64_t* data, uint32_t log_n, uint32_t batch, uint64_t stride,
                  int32_t inverse);
/* coset_fft / coset_ifft: forward evaluates on shift*<omega>; inverse interpolates from it. */
int32_t gl355_coset_ntt(gl355_ctx* ctx, uint64_t* data, uint32_t log_n, uint32_t batch,
                        uint64_t stride, uint64_t shift, int32_t inverse);

/* ---- a3: PolynomialCoeffs::lde + coset_fft_with_options --------------------------------------
 * coeffs: batch columns of n; out: batch columns of N = n << rate_bits;
 * natural order: out[c][j] = P_c(shift * omega_N^j). */
int32_t gl355_lde(gl355_ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits,
                  uint64_t shift, uint32_t batch, uint64_t* out);
/* same values in the commitment's bit-reversed order: out[c][i] = P_c(shift*omega_N^bitrev(i))
 * (fri_chip.rs:245-264).  This is the single-write fast path the commit pipeline uses. */
int32_t gl355_lde_bitrev(gl355_ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits,
                         uint64_t shift, uint32_t batch, uint64_t* out);

/* ---- a5: plonky2_util::transpose / reverse_index_bits_in_place (fri_chip.rs:6,189) ---------- */
/* out[c][r] = in[r][c]; in is rows x cols row-major */
int32_t gl355_transpose(gl355_ctx* ctx, const uint64_t* in, uint64_t rows, uint64_t cols, uint64_t* out);
/* rows of row_len u64: row i <-> row bitrev(i), in place */
int32_t gl355_reverse_index_bits(gl355_ctx* ctx, uint64_t* data, uint64_t n_rows, uint32_t row_len);

/* ---- a6: PoseidonPermutation::permute (access_set.rs:67; gates/poseidon.rs:26-322) ----------- */
int32_t gl355_poseidon_permute(gl355_ctx* ctx, uint64_t* states /* count x 12 */, uint64_t count);

/* ---- a7: hash_n_to_m_no_pad / hash_or_noop / two_to_one (hasher_chip.rs:122-148) ------------- */
/* n inputs of len elements each (row-major) -> n digests of 4; always runs the sponge */
int32_t gl355_hash_no_pad(gl355_ctx* ctx, const uint64_t* inputs, uint64_t n, uint32_t len, uint64_t* digests);
/* Merkle leaf digests: len <= 4 copies (zero padded), else sponge (merkle_proof_chip.rs:52-57) */
int32_t gl355_hash_leaves(gl355_ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len,
                          uint64_t* digests);
int32_t gl355_two_to_one(gl355_ctx* ctx, const uint64_t* left, const uint64_t* right, uint64_t n, uint64_t* out);

/* ---- a8: MerkleTree::new / prove (signal.rs:40, access_set.rs:205, recursion.rs:360, circuit.rs:91)
 * leaves: n_leaves x leaf_len row-major.  digests: 2*(n_leaves - 2^cap_height) x 4 u64 in
 * plonky2's layout (per cap subtree: left-subtree || left-child || right-child || right-subtree,
 * recursively) so MerkleTree::prove keeps working on it.  cap: 2^cap_height x 4. */
int32_t gl355_merkle_build(gl355_ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len,
                           uint32_t cap_height, uint64_t* digests, uint64_t* cap);
/* ---- the same entry points generic in the hasher `H` of plonky2's `Hasher<F>` / `GenericConfig::Hasher`
 * (MerkleTree::new::<F, H>, H::hash_no_pad, H::two_to_one, H::Permutation::permute).  GL355_HASH_POSEIDON is
 * PoseidonHash (the functions above); GL355_HASH_BN254_POSEIDON is the reference's Bn254PoseidonHash
 * (src/plonky2_verifier/bn245_poseidon/plonky2_config.rs:38-75 over native.rs:16-77: three Goldilocks elements per BN254
 * scalar, Poseidon t = 5 with 8 + 60 rounds and x^5, parameters constants.rs:5-404) -- the hasher of the outer wrap
 * proof (wrapper.rs:35-56, access_set.rs:48-49, recursion.rs:333-335).  Same argument meaning and digest layout. */
enum { GL355_HASH_POSEIDON = 0, GL355_HASH_BN254_POSEIDON = 1 };
int32_t gl355_permute_h(gl355_ctx* ctx, int32_t hasher, uint64_t* states /* count x 12 */, uint64_t count);
int32_t gl355_hash_no_pad_h(gl355_ctx* ctx, int32_t hasher, const uint64_t* inputs, uint64_t n, uint32_t len, uint64_t* digests);
int32_t gl355_hash_leaves_h(gl355_ctx* ctx, int32_t hasher, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len,
                            uint64_t* digests);
int32_t gl355_two_to_one_h(gl355_ctx* ctx, int32_t hasher, const uint64_t* left, const uint64_t* right, uint64_t n, uint64_t* out);
int32_t gl355_merkle_build_h(gl355_ctx* ctx, int32_t hasher, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len,
                             uint32_t cap_height, uint64_t* digests, uint64_t* cap);
/* siblings leaf -> cap: (log2(n_leaves) - cap_height) x 4 u64 (host buffer) */
int32_t gl355_merkle_prove(gl355_ctx* ctx, const uint64_t* digests, uint64_t n_leaves, uint32_t cap_height,
                           uint64_t leaf_index, uint64_t* siblings);

/* ---- a4 (+a14): PolynomialBatch::from_values / from_coeffs (4x per proof) ---------------------
 * values: batch columns of n (column-major), evaluations on <omega_n> (is_coeffs = 0) or
 * coefficients (is_coeffs = 1).  salt: NULL, or GL355_SALT_SIZE columns of N = n << rate_bits
 * random elements appended to every leaf when the oracle is blinded (natural order, i.e. exactly
 * what plonky2 appends to the LDE before its transpose).  The oracle keeps coefficients, the LDE
 * (column-major, bit-reversed row order), digests and cap resident on the device. */
/* gl355_commit_h: the same with the Merkle hasher of the configuration (GL355_HASH_*); gl355_commit = GL355_HASH_POSEIDON */
int32_t gl355_commit_h(gl355_ctx* ctx, int32_t hasher, const uint64_t* values, uint32_t log_n, uint32_t batch, uint32_t rate_bits,
                       int32_t is_coeffs, const uint64_t* salt, uint32_t cap_height, gl355_oracle** out);
int32_t gl355_commit(gl355_ctx* ctx, const uint64_t* values, uint32_t log_n, uint32_t batch,
                     uint32_t rate_bits, int32_t is_coeffs, const uint64_t* salt, uint32_t cap_height,
                     gl355_oracle** out);
int32_t gl355_oracle_destroy(gl355_oracle* o);
int32_t gl355_oracle_info(const gl355_oracle* o, uint32_t* log_n, uint32_t* rate_bits, uint32_t* batch,
                          uint32_t* leaf_len, uint32_t* cap_height);
int32_t gl355_oracle_cap(const gl355_oracle* o, uint64_t* cap /* 2^cap_height x 4 */);
int32_t gl355_oracle_coeffs(const gl355_oracle* o, uint64_t* coeffs /* batch x n */);
/* plonky2-layout exports (row-major leaves [N][leaf_len], digests) for MerkleTree { leaves, digests, cap } */
int32_t gl355_oracle_leaves(const gl355_oracle* o, uint64_t* leaves);
int32_t gl355_oracle_digests(const gl355_oracle* o, uint64_t* digests);
/* device pointers for resident pipelines: LDE column c at lde + c*N (bit-reversed rows) */
const uint64_t* gl355_oracle_lde_ptr(const gl355_oracle* o);
const uint64_t* gl355_oracle_coeffs_ptr(const gl355_oracle* o);
/* MerkleTree::get + prove for one query index (fri_prover_query_round): leaf [leaf_len] and
 * siblings [(log2 N - cap_height) x 4] to host buffers */
int32_t gl355_oracle_open(const gl355_oracle* o, uint64_t index, uint64_t* leaf, uint64_t* siblings);

/* all FRI queries of one oracle at once: leaves[n_idx][leaf_len], siblings[n_idx][log2 N - cap_height][4] (host) */
int32_t gl355_oracle_open_batch(const gl355_oracle* o, const uint64_t* indices, uint32_t n_idx, uint64_t* leaves,
                                uint64_t* siblings);

/* ---- a10: compute_quotient_polys (vanishing_poly.rs:18-153, gates/mod.rs:87-132, gates/ evaluators) ------
 * Circuit shape the constraint kernel needs (the host-side CommonCircuitData, types/common_data.rs:69-97).
 * gates[] is the circuit's gate list in selector order; gate i is active on rows where
 * constants[selector_index](row) == i; group = [group_start, group_end) are the gate indices sharing
 * that selector polynomial. */
#define GL355_MAX_GATES 16
enum { GL355_GATE_NOOP = 0, GL355_GATE_CONSTANT = 1, GL355_GATE_PUBLIC_INPUT = 2, GL355_GATE_BASE_SUM = 3,
       GL355_GATE_POSEIDON = 4, GL355_GATE_ARITHMETIC = 5, GL355_GATE_ARITHMETIC_EXT = 6, GL355_GATE_MUL_EXT = 7,
       GL355_GATE_POSEIDON_MDS = 8, GL355_GATE_RANDOM_ACCESS = 9, GL355_GATE_REDUCING = 10,
       GL355_GATE_REDUCING_EXT = 11 };   /* the gate set of gates/mod.rs:141-196 */
#define GL355_GATE_TYPE_MAX GL355_GATE_REDUCING_EXT
/* RANDOM_ACCESS param = bits | num_copies << 8 | num_extra_constants << 16 */
typedef struct {
    uint32_t type;            /* GL355_GATE_* */
    uint32_t param;           /* CONSTANT: num_consts; BASE_SUM: num_limbs (base 2); ARITHMETIC(_EXT) and MUL_EXT: num_ops;
                                 REDUCING*: num_coeffs; RANDOM_ACCESS: packed (above) */
    uint32_t selector_index;  /* which constants column is this gate's selector */
    uint32_t group_start, group_end;
} gl355_gate;
typedef struct {
    uint32_t degree_bits, rate_bits;
    uint32_t num_wires, num_routed_wires;
    uint32_t num_constants;         /* gate constants (columns after the selectors) */
    uint32_t num_selectors;
    uint32_t num_challenges;
    uint32_t max_degree;            /* quotient_degree_factor = partial-product chunk size (8) */
    uint32_t num_partial_products;  /* ceil(routed / max_degree) - 1 */
    uint32_t num_gates;
    gl355_gate gates[GL355_MAX_GATES];
} gl355_circuit;
/* Evaluates the combined vanishing polynomial / Z_H on the quotient coset for every challenge,
 * interpolates (coset iNTT) and writes the quotient chunks: out[(c*max_degree + j)][n] coefficients.
 * Oracles: constants_sigmas = [selectors | gate constants | sigmas], wires, zs_partial_products =
 * [Z_c]_c | [pp_{c,k}]_{c,k}.  k_is[num_routed_wires], betas/gammas/alphas[num_challenges] base field. */
int32_t gl355_quotient(gl355_ctx* ctx, const gl355_circuit* circuit, const gl355_oracle* constants_sigmas,
                       const gl355_oracle* wires, const gl355_oracle* zs_partial_products, const uint64_t* k_is,
                       const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas,
                       const uint64_t pi_hash[4], uint64_t* quotient_coeffs);
/* the vanishing values themselves (before interpolation), storage order = bit-reversed rows of the
 * quotient coset: values[c][t]; for parity tests */
int32_t gl355_quotient_values(gl355_ctx* ctx, const gl355_circuit* circuit, const gl355_oracle* constants_sigmas,
                              const gl355_oracle* wires, const gl355_oracle* zs_partial_products, const uint64_t* k_is,
                              const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas,
                              const uint64_t pi_hash[4], uint64_t* values);

/* ---- a11: PolynomialBatch::prove_openings (fri_chip.rs:112-149) ------------------------------
 * One opening batch over polynomials taken from resident oracles:
 *   C(X) = sum_i alpha^i p_i(X);  Q = (C - C(z)) / (X - z), padded to n;  acc = acc*alpha^k + Q
 * polys[i] = (oracle, column) in batch order; acc: n extension coefficients (in/out). */
typedef struct { const gl355_oracle* oracle; uint32_t column; } gl355_poly_ref;
int32_t gl355_deep_batch(gl355_ctx* ctx, const gl355_poly_ref* polys, uint32_t n_polys,
                         const uint64_t alpha[2], const uint64_t z[2], uint64_t* acc);
/* openings: out[i] = p_i(z) in the extension field (OpeningSet::new) */
int32_t gl355_eval_polys(gl355_ctx* ctx, const gl355_poly_ref* polys, uint32_t n_polys,
                         const uint64_t z[2], uint64_t* out);
/* extension-field LDE of the final DEEP polynomial: coeffs n ext -> out N ext, natural order */
int32_t gl355_lde_ext(gl355_ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits,
                      uint64_t shift, uint64_t* out);

/* ---- a12: fri_committed_trees (fri_chip.rs:168-226,275-316) ---------------------------------- */
/* arity-2 fold of n extension coefficients: out[k] = c[2k] + beta*c[2k+1] */
int32_t gl355_fri_fold(gl355_ctx* ctx, const uint64_t* coeffs, uint64_t n, const uint64_t beta[2], uint64_t* out);
/* commit-phase layer tree from natural-order extension values (n ext): leaves = pairs in
 * bit-reversed order, flattened to 4 u64 (no leaf hash); outputs as gl355_merkle_build */
int32_t gl355_fri_layer_commit_h(gl355_ctx* ctx, int32_t hasher, const uint64_t* values, uint64_t n, uint32_t cap_height,
                                 uint64_t* leaves, uint64_t* digests, uint64_t* cap);
int32_t gl355_fri_layer_commit(gl355_ctx* ctx, const uint64_t* values, uint64_t n, uint32_t cap_height,
                               uint64_t* leaves, uint64_t* digests, uint64_t* cap);

/* ---- a13: fri_proof_of_work (fri_chip.rs:364-376, plonk_verifier_chip.rs:136-137) ------------- */
/* smallest w >= start such that permute(state with state[pos] = w)[7] has >= bits leading zeros */
int32_t gl355_pow_grind_h(gl355_ctx* ctx, int32_t hasher, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start,
                          uint64_t* witness);
int32_t gl355_pow_grind(gl355_ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits,
                        uint64_t start, uint64_t* witness);

/* ---- a15: Challenger (host, sequential; plonk_verifier_chip.rs:55-154, hasher_chip.rs:48-89) ----
 * Plain host struct, no device involved: observe buffers inputs (absorbed 8 at a time by OVERWRITING
 * the state), squeeze pops from the end of the rate part. */
typedef struct {
    uint64_t state[12];
    uint64_t in_buf[8];
    uint32_t in_len;
    uint64_t out_buf[8];
    uint32_t out_len;
    int32_t hasher;          /* GL355_HASH_*: the sponge permutation (GenericConfig::Hasher); 0 after gl355_challenger_init */
} gl355_challenger;
int32_t gl355_challenger_init(gl355_challenger* c);
int32_t gl355_challenger_init_h(gl355_challenger* c, int32_t hasher);
int32_t gl355_challenger_observe(gl355_challenger* c, const uint64_t* elems, uint64_t n);
int32_t gl355_challenger_squeeze(gl355_challenger* c, uint64_t* out, uint64_t n);
/* sponge state + slot of the PoW witness candidate, to feed gl355_pow_grind */
int32_t gl355_challenger_pow_state(const gl355_challenger* c, uint64_t state[12], uint32_t* pos);
/* single host-side hashes (access_set.rs:67 hashes one 8-element input; the transcript) */
int32_t gl355_host_poseidon_permute(uint64_t state[12]);
int32_t gl355_host_hash_no_pad(const uint64_t* in, uint64_t len, uint64_t out[4]);
int32_t gl355_host_permute_h(int32_t hasher, uint64_t state[12]);
int32_t gl355_host_hash_no_pad_h(int32_t hasher, const uint64_t* in, uint64_t len, uint64_t out[4]);
/* witness generation of one PoseidonGate row (wire layout gates/poseidon.rs:329-380) */
int32_t gl355_poseidon_gate_witness(const uint64_t inputs[12], uint64_t swap, uint64_t wires[135]);

/* ---- a12 + a13 + a14 in one call: fri_proof (fri_chip.rs:168-226,275-327,364-376) ------------------
 * final_coeffs: n = 2^log_n extension coefficients of the DEEP polynomial (host or device).  Runs the
 * commit phase (n_layers arity-2 folds), observes the final polynomial, grinds the PoW, squeezes the
 * query indices and opens every layer tree at every query; `ch` is advanced exactly as
 * plonk_verifier_chip.rs:120-140 replays it.  Host outputs:
 *   caps[n_layers][2^cap_height][4], final_poly[(n >> n_layers) ext], pow_witness, query_indices[num_queries],
 *   step_evals[num_queries][n_layers][4], step_siblings[num_queries][sum_l (log_n+rate_bits-1-l-cap_height)][4]. */
int32_t gl355_fri_prove(gl355_ctx* ctx, const uint64_t* final_coeffs, uint32_t log_n, uint32_t rate_bits,
                        uint32_t cap_height, const uint32_t* arity_bits, uint32_t n_layers, uint32_t pow_bits,
                        uint32_t num_queries, gl355_challenger* ch, uint64_t* caps, uint64_t* final_poly,
                        uint64_t* pow_witness, uint64_t* query_indices, uint64_t* step_evals, uint64_t* step_siblings);

/* ---- CircuitData::prove in one call (access_set.rs:94, recursion.rs:168, wrapper.rs:55) ----------------
 * Prover data of one built circuit (plonky2 ProverOnlyCircuitData + CommonCircuitData, the parts the hot
 * path needs).  constants_sigmas is the preprocessed oracle committed at build time (not blinded);
 * sigmas are the sigma VALUES sigma[j][i] = k_{j'} * g^{i'} ([num_routed_wires][n], host or device). */
typedef struct {
    const gl355_circuit* circuit;
    const gl355_oracle* constants_sigmas;
    const uint64_t* sigmas;
    const uint64_t* k_is;
    uint64_t circuit_digest[4];
    uint32_t cap_height, pow_bits, num_queries, n_fri_layers;
    int32_t zero_knowledge;      /* salt the wires / Z / quotient oracles with 4 pseudo-random columns */
    int32_t hasher;              /* GL355_HASH_*: GenericConfig::Hasher = Merkle trees, transcript and PoW of this proof
                                    (constants_sigmas must have been committed with it, gl355_commit_h).  Public inputs are
                                    always hashed with Poseidon (GenericConfig::InnerHasher). */
} gl355_prover_data;
/* u64 words of the flat proof gl355_prove writes */
uint64_t gl355_proof_words(const gl355_prover_data* pd);
/* Zero-knowledge blinding (zero_knowledge: true at access_set.rs:69, recursion.rs:33; SALT_SIZE types/assigned.rs:67-71).
 * Every proving entry point takes `blinding_key`: 32 bytes, or NULL.
 *   NULL      the library draws a fresh 256-bit key from the OS CSPRNG (getrandom) for this proof -- the production setting,
 *             the counterpart of plonky2's OsRng;
 *   non-NULL  the caller's SECRET key: a (witness, key) pair then gives a reproducible proof (tests, the byte-for-byte parity
 *             with the CPU restatement, deterministic provers that derive the key from their own secret).  A key must never be
 *             reused for a different witness and must be as secret as the witness: it determines every blinding value.
 * Salt columns and blinding rows are ChaCha20 key streams (RFC 8439 block function, counter = block index, nonce = (stream, 0, 0);
 * streams 1/2/3 = salt of the wires / Z / quotient oracle, 4 = witness blinding rows); stream element k = key-stream bytes
 * [16k, 16k+16) as a little-endian 128-bit number reduced mod p.  The published salt therefore reveals nothing about the key or
 * about the wire blinding. */
int32_t gl355_derive_key(const uint8_t base_key[32], uint64_t index, uint8_t out[32]);  /* per-unit key of a batch: ChaCha20(base, nonce ("key", index)) */
/* the first `count` elements of a blinding stream, from the device kernel the prover uses (parity surface) */
int32_t gl355_blinding_elements(gl355_ctx* ctx, const uint8_t key[32], uint32_t stream, uint64_t count, uint64_t* out);

/* wires: the full witness [num_wires][n] (host or device), including the blinding rows; blinding_key drives the salt columns.
 * Flat proof layout (all u64; E = extension element = 2 words, H = digest = 4 words, C = 2^cap_height):
 *   header[8] = {total_words, degree_bits, n_fri_layers, num_queries, n_public_inputs, zero_knowledge, cap_height, num_challenges}
 *   wires_cap[C]H  zs_partial_products_cap[C]H  quotient_polys_cap[C]H
 *   openings at zeta: constants | sigmas | wires | zs | partial_products | quotient_polys (E each), then zs at g*zeta
 *   fri caps[n_fri_layers][C]H   final_poly[n >> n_fri_layers]E   pow_witness
 *   per query: x_index; for oracle 0..3: leaf[leaf_len], siblings[log2 N - cap_height]H;
 *              for layer l: evals[2]E, siblings[log2 N - 1 - l - cap_height]H */
int32_t gl355_prove(gl355_ctx* ctx, const gl355_prover_data* pd, const uint64_t* wires, const uint64_t* public_inputs,
                    uint32_t n_public_inputs, const uint8_t* blinding_key, uint64_t* proof, uint64_t proof_capacity_words);

/* the same proof from a SPARSE witness: only the n_rows non-trivial circuit rows are given (rows[r] = all
 * num_wires values of circuit row row_idx[r]; all other rows are zero Noop rows), and the zero-knowledge
 * blinding rows are filled on the device from stream 4 of `blinding_key`: rows [blind_start, blind_start + n_blind) random on
 * every wire, n_z_pairs consecutive row pairs from z_start carrying one random value per routed wire, shared by the two rows of
 * the pair (plonky2 `blind`; the circuit must copy-constrain every routed column between the rows of a pair). */
int32_t gl355_prove_sparse(gl355_ctx* ctx, const gl355_prover_data* pd, const uint32_t* row_idx, const uint64_t* rows,
                           uint32_t n_rows, uint32_t blind_start, uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs,
                           const uint64_t* public_inputs, uint32_t n_public_inputs, const uint8_t* blinding_key, uint64_t* proof,
                           uint64_t proof_capacity_words);
/* n_units (<= GL355_MAX_UNITS) independent witnesses of ONE circuit proven in lock-step on this context -- the reference's
 * rayon par_iter over proofs (recursion.rs:214-227,300-308) as a unit dimension of every kernel, which is what fills the GPU with
 * circuits of n = 2^13..2^15: rows = [n_units][n_rows][num_wires], public_inputs = [n_units][n_public_inputs], proofs =
 * [n_units][proof_capacity_words], blinding_keys = [n_units][32] or NULL (a fresh OS-random key for every unit).  Every unit's
 * proof is byte for byte the proof gl355_prove_sparse makes of that unit alone with the same key. */
int32_t gl355_prove_sparse_units(gl355_ctx* ctx, const gl355_prover_data* pd, uint32_t n_units, const uint32_t* row_idx, const uint64_t* rows,
                                 uint32_t n_rows, uint32_t blind_start, uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs,
                                 const uint64_t* public_inputs, uint32_t n_public_inputs, const uint8_t* blinding_keys,
                                 uint64_t* proofs, uint64_t proof_capacity_words);
/* ---- CircuitData::verify (access_set.rs:170-175: `self.verifier_data.verify(signal.proof)`) ------------------------------------
 * Host-only check of a flat proof (the layout above) against the verifier data of its circuit: transcript replay, the vanishing
 * identity at zeta with every gate evaluator, proof of work, and every FRI query (initial-tree and layer Merkle paths, batch
 * combination, folds, final polynomial) -- the equations of chip/plonk/plonk_verifier_chip.rs:55-242, vanishing_poly.rs:18-218,
 * fri_chip.rs:58-376, merkle_proof_chip.rs:39-87.  GL355_OK = valid; GL355_E_VERIFY = rejected (gl355_verify_last_error says which
 * check); GL355_E_INVALID_ARG = unusable verifier data. */
typedef struct {
    const gl355_circuit* circuit;
    const uint64_t* constants_sigmas_cap;   /* [2^cap_height][4]: the preprocessed commitment (VerifierOnlyCircuitData) */
    const uint64_t* k_is;                   /* [num_routed_wires] */
    uint64_t circuit_digest[4];
    uint32_t cap_height, pow_bits, num_queries, n_fri_layers;
    int32_t zero_knowledge, hasher;
} gl355_verifier_data;
int32_t gl355_verify(const gl355_verifier_data* vd, const uint64_t* proof, uint64_t proof_words, const uint64_t* public_inputs,
                     uint32_t n_public_inputs);
const char* gl355_verify_last_error(void);   /* of the calling thread's last gl355_verify / gl355_circuit_verify */

/* host-side witness rows of the Semaphore circuit (circuit.rs:67-99): (height + 7) rows x 135 wires in the
 * order PublicInput | pi-hash 1 | pi-hash 2 | BaseSum{height} | leaf hash | height Merkle levels | nullifier |
 * Constant; public_inputs = merkle_root | nullifier | topic (circuit.rs:27-32). */
int32_t gl355_semaphore_witness(const uint64_t private_key[4], const uint64_t topic[4], uint64_t index,
                                const uint64_t* siblings, uint32_t height, uint64_t* rows, uint64_t public_inputs[12]);

/* Witness tape: the host-side witness generation of a built circuit (plonky2's generators, run inside
 * `data.prove(pw)` at recursion.rs:167-168 / wrapper.rs:55) as a recorded straight-line program.  tape is
 * n_ops entries of 5 words {op, a, b, c, d}; wire operands are positions in `rows` (sparse row r, wire w ->
 * r * num_wires + w, the layout gl355_prove_sparse takes); `inputs` is the flat input vector the INPUT
 * entries index (for the recursive verifier: each inner proof's flat words followed by its public inputs).
 *   CONST a <- b | INPUT a <- inputs[b] | COPY a <- [b] | ASSERT_EQ [a] == [b]
 *   ARITH at a: {m0, m1, addend, out}, out = b*m0*m1 + c*addend            (gates/arithmetic.rs)
 *   ARITH_EXT at a: the same over F_p^2, 8 wires                           (gates/arithmetic_extension.rs)
 *   POSEIDON row a: inputs 0..11 and swap 24 set, fills the row            (gates/poseidon.rs:329-380)
 *   MDS_EXT row a: 12 extension inputs -> 12 outputs at wire 24            (gates/poseidon_mds.rs)
 *   BASE_SUM row a: wire 0 -> b little-endian bits at wire 1               (gates/base_sum.rs)
 *   RANDOM_ACCESS row a, copy b: claimed element + index bits              (gates/random_access.rs)
 *   REDUCING row a: b coefficients, c = 1 for the extension variant        (gates/reducing.rs, reducing_extension.rs)
 *   LO32 / HI32 a <- halves of [b] | EXT_INV (a, b) <- ([c], [d])^-1
 * rows is zeroed first.  Returns GL355_E_WITNESS (and the entry index in *failed_op) when an ASSERT_EQ or a
 * range condition fails, i.e. the inputs do not satisfy the circuit. */
enum { GL355_TAPE_CONST = 0, GL355_TAPE_INPUT = 1, GL355_TAPE_COPY = 2, GL355_TAPE_ASSERT_EQ = 3, GL355_TAPE_ARITH = 4,
       GL355_TAPE_ARITH_EXT = 5, GL355_TAPE_POSEIDON = 6, GL355_TAPE_MDS_EXT = 7, GL355_TAPE_BASE_SUM = 8,
       GL355_TAPE_RANDOM_ACCESS = 9, GL355_TAPE_REDUCING = 10, GL355_TAPE_LO32 = 11, GL355_TAPE_HI32 = 12,
       GL355_TAPE_EXT_INV = 13 };
int32_t gl355_witness_replay(const uint64_t* tape, uint64_t n_ops, const uint64_t* inputs, uint64_t n_inputs,
                             uint64_t* rows, uint64_t n_words, uint32_t num_wires, uint64_t* failed_op);
/* A segmented tape = [n_seq sequential entries | segment 0 | segment 1 | ...] where the builder has checked that a segment reads
 * only the sequential part and itself (the FRI query rounds of a verifier circuit): the segments run on `threads` host threads.
 * Same rows and same error reporting (smallest failing entry) as the sequential replay of the same tape. */
int32_t gl355_witness_replay_segmented(const uint64_t* tape, uint64_t n_ops, uint64_t n_seq, const uint64_t* seg_lens, uint32_t n_segs,
                                       uint32_t threads, const uint64_t* inputs, uint64_t n_inputs, uint64_t* rows, uint64_t n_words,
                                       uint32_t num_wires, uint64_t* failed_op);

/* ---- circuit artifacts: the native per-proof path --------------------------------------------------------------
 * A circuit is BUILT once (plonky2's CircuitBuilder::build at access_set.rs:91, recursion.rs:167, wrapper.rs:41; here the
 * host-side builder stark-verifier_amd/plonk.py) and serialised with CircuitData.export_blob(): u64 words
 *   [0] magic "GL355CIR" [1] version 2 (3: the digest at [106..109] was computed elsewhere, e.g. by plonky2's CircuitBuilder::build,
 *   and the artifact ends with the expected constants_sigmas cap [2^cap_height][4], which is checked instead of the digest) [2..11] the gl355_circuit scalars [12..91] 16 gates x {type, param, selector_index,
 *   group_start, group_end} [92] cap_height [93] pow_bits [94] num_queries [95] n_fri_layers [96] zero_knowledge [97] hasher
 *   [98] blind_start [99] n_blind [100] z_start [101] n_z_pairs [102] n_rows [103] n_tape_ops [104] n_inputs [105] n_public_inputs
 *   [106..109] circuit digest [110] sequential tape entries [111] independent tape segments, then constants[num_selectors + num_constants][n] | sigmas[routed][n] | k_is[routed] |
 *   row_idx[n_rows] | public-input positions[n_pi] | tape[n_tape_ops][5] (gl355_witness_replay format) | segment lengths.
 * gl355_circuit_load commits constants_sigmas on `ctx`, re-derives the circuit digest and refuses an artifact whose digest
 * does not match its tables.  The handle is read-only afterwards: any context of the same device may prove with it, concurrently.
 *   gl355_semaphore_prove    = fill_semaphore_targets + data.prove (access_set.rs:61-104): witness rows from the member's
 *                              key / topic / Merkle path, proof, public inputs root | nullifier | topic
 *   gl355_circuit_prove_tape = set_proof_with_pis_target + data.prove (recursion.rs:72-86,167-168, wrapper.rs:49-55): inputs =
 *                              each inner proof's flat words followed by its public inputs; GL355_E_WITNESS on an invalid inner proof
 *   gl355_circuit_prove_rows = data.prove from ready-made sparse rows (in the artifact's row order) */
typedef struct gl355_circuit_handle gl355_circuit_handle;
int32_t gl355_circuit_load(gl355_ctx* ctx, const uint64_t* blob, uint64_t words, gl355_circuit_handle** out);
int32_t gl355_circuit_destroy(gl355_circuit_handle* c);
int32_t gl355_circuit_info(const gl355_circuit_handle* c, uint64_t* proof_words, uint32_t* n_public_inputs, uint32_t* n_rows,
                           uint64_t* n_inputs, uint32_t* degree_bits);
const uint64_t* gl355_circuit_digest(const gl355_circuit_handle* c);
/* the witness rows [n_units][n_rows][num_wires] and public inputs [n_units][n_public_inputs] the circuit's tape generates from
 * `inputs` [n_units][n_inputs] -- on host threads (on_device = 0) or by the device interpreter (on_device = 1); both must agree
 * (parity surface of SURVEY 8(f) N3).  GL355_E_WITNESS + *failed_entry when the inputs do not satisfy the circuit. */
int32_t gl355_circuit_witness_rows(gl355_ctx* ctx, const gl355_circuit_handle* c, uint32_t n_units, const uint64_t* inputs, uint64_t n_inputs,
                                   int32_t on_device, uint64_t* rows, uint64_t* public_inputs_out, uint64_t* failed_entry);
/* gl355_verify with the verifier data of a loaded circuit */
int32_t gl355_circuit_verify(const gl355_circuit_handle* c, const uint64_t* proof, uint64_t proof_words, const uint64_t* public_inputs,
                             uint32_t n_public_inputs);
int32_t gl355_circuit_prove_rows(gl355_ctx* ctx, const gl355_circuit_handle* c, const uint64_t* rows, const uint64_t* public_inputs,
                                 uint32_t n_public_inputs, const uint8_t* blinding_key, uint64_t* proof, uint64_t proof_capacity_words);
int32_t gl355_circuit_prove_tape(gl355_ctx* ctx, const gl355_circuit_handle* c, const uint64_t* inputs, uint64_t n_inputs, const uint8_t* blinding_key,
                                 uint64_t* proof, uint64_t proof_capacity_words, uint64_t* public_inputs_out);
int32_t gl355_semaphore_prove(gl355_ctx* ctx, const gl355_circuit_handle* c, const uint64_t private_key[4], const uint64_t topic[4],
                              uint64_t index, const uint64_t* siblings, uint32_t height, const uint8_t* blinding_key, uint64_t* proof,
                              uint64_t proof_capacity_words, uint64_t public_inputs_out[12]);

/* the same three calls for n_units (<= GL355_MAX_UNITS) units in lock-step (gl355_prove_sparse_units): rows [n_units][n_rows][num_wires],
 * public_inputs [n_units][n_public_inputs], inputs [n_units][n_inputs], private_keys [n_units][4], topics [n_units][4], indices [n_units],
 * siblings [n_units][height][4], blinding_keys [n_units][32] or NULL, proofs [n_units][proof_words], public_inputs_out [n_units][n_public_inputs].
 * gl355_circuit_prove_tape_units replays the units' tapes on GL355_OPT_REPLAY_THREADS host threads. */
int32_t gl355_circuit_prove_rows_units(gl355_ctx* ctx, const gl355_circuit_handle* c, uint32_t n_units, const uint64_t* rows, const uint64_t* public_inputs,
                                       uint32_t n_public_inputs, const uint8_t* blinding_keys, uint64_t* proofs);
int32_t gl355_circuit_prove_tape_units(gl355_ctx* ctx, const gl355_circuit_handle* c, uint32_t n_units, const uint64_t* inputs, uint64_t n_inputs,
                                       const uint8_t* blinding_keys, uint64_t* proofs, uint64_t* public_inputs_out);
int32_t gl355_semaphore_prove_units(gl355_ctx* ctx, const gl355_circuit_handle* c, uint32_t n_units, const uint64_t* private_keys, const uint64_t* topics,
                                    const uint64_t* indices, const uint64_t* siblings, uint32_t height, const uint8_t* blinding_keys, uint64_t* proofs,
                                    uint64_t* public_inputs_out);

/* Batch runtime (recursion.rs:300-308 `par_iter` of make_signal, :211-227 of the verification circuits): one host thread per
 * context; a context takes the next GL355_OPT_BATCH_UNITS units at a time and proves them in lock-step (results are placed by j and
 * every unit has its own key, so they do not depend on which context proved which unit, nor on the batch size).  Per unit: Merkle path of member_indices[j] from tree_digests (the access-set tree over
 * the public keys, cap height 0, plonky2 digest layout, host memory), gl355_semaphore_prove with key gl355_derive_key(key_base, 2j), and
 * if `rec` is not NULL gl355_circuit_prove_tape(rec, proof | public inputs) with key gl355_derive_key(key_base, 2j + 1); key_base NULL =
 * a fresh OS-random key per proof.  leaves_out[j] = nullifier | topic
 * (8 words) of unit j; proofs_out (optional) receives the last proof of every unit; units_per_ctx (optional) the units each
 * context proved.  Returns the first error (the failing context's gl355_last_error tells more). */
int32_t gl355_semaphore_units(gl355_ctx* const* ctxs, uint32_t n_ctx, const gl355_circuit_handle* sem, const gl355_circuit_handle* rec,
                              const uint64_t* private_keys, uint64_t n_members, const uint64_t topic[4], const uint64_t* tree_digests,
                              const uint64_t* member_indices, uint32_t count, const uint8_t* key_base, uint64_t* leaves_out,
                              uint64_t* proofs_out, uint32_t* units_per_ctx);

/* recursion.rs:187-247 `aggregate` as ONE native call: the binary aggregation tree over n_leaves = 2^n_levels proofs of one circuit (the
 * reference's benchmark flow, README.md:167-177).  levels[l] = the circuit verifying two proofs of tree level l (level 0: the leaves), each
 * loaded once from its artifact (the reference rebuilds it inside every aggregate_signals call, recursion.rs:25-185).  The nodes of a level are
 * independent (par_chunks_exact(2), recursion.rs:211-227): context t proves nodes t, t + n_ctx, ... in lock-step batches; levels are separated by a
 * join.  Node j of level l is blinded with gl355_derive_key(key_base, key_domain << 48 | l << 32 | j) (key_base NULL: fresh OS randomness per proof),
 * so the root does not depend on the number of contexts.  leaf_proofs [n_leaves][leaf_words], leaf_public_inputs [n_leaves][leaf_n_pi] (host);
 * proof_out / public_inputs_out receive the root proof (gl355_circuit_info(levels[n_levels - 1]) gives the sizes); level_ms (optional,
 * n_levels doubles): wall milliseconds per level.  GL355_E_WITNESS if an inner proof does not verify. */
int32_t gl355_aggregate_units(gl355_ctx* const* ctxs, uint32_t n_ctx, const gl355_circuit_handle* const* levels, uint32_t n_levels,
                              const uint64_t* leaf_proofs, const uint64_t* leaf_public_inputs, uint32_t n_leaves, uint64_t leaf_words, uint32_t leaf_n_pi,
                              const uint8_t* key_base, uint64_t key_domain, uint64_t* proof_out, uint64_t proof_capacity_words,
                              uint64_t* public_inputs_out, uint32_t public_inputs_capacity, double* level_ms);

/* ---- multi-GPU exchange: the counterpart of the reference's collection of its parallel proofs into one Vec
 * (recursion.rs:189-227 `Mutex<Vec<..>>` behind par_chunks_exact, :300-308 par_iter().collect()).  One process per GPU; the units
 * of a batch are block-partitioned over the ranks with no data-path collective; the only exchange is this all-gather of one small
 * leaf per unit (nullifier | topic, recursion.rs:110-165) or of one proof per rank, then rank 0 builds the aggregation root.
 *   GL355_COMM_RCCL  ncclAllGather / ncclAllReduce over xGMI on the context's stream; librccl is bound on first use.  The
 *                    communicator is created from a unique id the caller distributes: rank 0 calls gl355_comm_unique_id and
 *                    hands the 128 bytes to the other ranks by the host's own means (launcher, file, socket).
 *   GL355_COMM_HOST  the same calls over TCP between the host processes, rank 0 listening on the address named by
 *                    gl355_comm_host_id (hosts without RCCL; CPU tests).  Selected explicitly, never a fallback.  Like an
 *                    ncclUniqueId the id is minted ONCE (rank 0) and handed to the other ranks: it carries a 128-bit random token
 *                    that every rank presents in its handshake; connections without it are closed and ignored, rank 0 keeps
 *                    accepting until all ranks have arrived (120 s), and the data sockets time out (GL355_COMM_TIMEOUT_S, 600 s)
 *                    instead of hanging on a dead peer.
 * gl355_gather_digests: every rank contributes words_per_rank u64 (host or device memory); `all` receives world * words_per_rank
 * words in rank order on every rank.  gl355_comm_barrier / gl355_comm_max_f64: what a benchmark or a driver needs around it. */
#define GL355_COMM_ID_BYTES 128
enum { GL355_COMM_RCCL = 0, GL355_COMM_HOST = 1 };
typedef struct gl355_comm gl355_comm;
int32_t gl355_comm_unique_id(int32_t backend, uint8_t id[GL355_COMM_ID_BYTES]);
int32_t gl355_comm_host_id(const char* ipv4, uint16_t port, uint8_t id[GL355_COMM_ID_BYTES]);
int32_t gl355_comm_create(gl355_ctx* ctx /* may be NULL for GL355_COMM_HOST */, int32_t backend, const uint8_t id[GL355_COMM_ID_BYTES],
                          int32_t rank, int32_t world, gl355_comm** out);
int32_t gl355_comm_destroy(gl355_comm* comm);
int32_t gl355_comm_info(const gl355_comm* comm, int32_t* rank, int32_t* world, int32_t* backend);
const char* gl355_comm_last_error(gl355_comm* comm);   /* comm == NULL: the last failed create / id call of this thread */
int32_t gl355_gather_digests(gl355_comm* comm, const uint64_t* local, uint64_t words_per_rank, uint64_t* all);
int32_t gl355_comm_barrier(gl355_comm* comm);
int32_t gl355_comm_max_f64(gl355_comm* comm, double* inout);
/* Poseidon-Goldilocks Merkle root (cap height 0) over n_leaves leaves of leaf_len words, zero-padded to a power of two:
 * MerkleTree::new(leaves, 0).cap[0], the "aggregation root" over the gathered (nullifier | topic) leaves */
int32_t gl355_aggregation_root(gl355_ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, uint64_t root[4]);

/* ---- SURVEY 8(f) N4, first slice: the two kernels of the Halo2 / KZG finalisation (verifier_api.rs:57-96: ParamsKZG::setup :77,
 * keygen_vk / keygen_pk :78-79, create_proof :90, at k = 20..23 per chip/native_chip/test_utils.rs:57-95) -- halo2_proofs'
 * `best_fft` over halo2curves bn256::Fr and `best_multiexp` over bn256::G1.  Values are canonical integers as 4 little-endian u64
 * (any 256-bit value is accepted and reduced); G1 points are affine x | y (8 u64), (0, 0) encodes the identity.
 *   gl355_bn254_fr_ntt   in place, natural order in and out: a[k] <- sum_i a[i] w^(ik), w = ROOT_OF_UNITY^(2^(28 - log_n));
 *                        inverse != 0: with w^-1 and the 1/n scaling (EvaluationDomain::ifft)
 *   gl355_bn254_fr_coset_ntt   EvaluationDomain::coeff_to_extended / extended_to_coeff: forward, 2^log_small coefficients c_i ->
 *                        out[k] = sum_i c_i shift^i w^(ik) for k < 2^log_n (zero-padded, w the 2^log_n-th root); inverse, 2^log_n evaluations ->
 *                        the first 2^log_small coefficients, each divided by shift^i (and by 2^log_n).  in != out
 *   gl355_bn254_g1_msm   result = sum_i scalars[i] * points[i]  (Pippenger buckets, signed digits; the windows are combined on the host)
 *   gl355_bn254_g1_msm_batch   n_sets MSMs over the SAME bases (the commitments of several columns under one SRS: ParamsKZG::commit per
 *                        advice / fixed / permutation polynomial): the sets' windows run as more windows of one MSM, so the point conversion is
 *                        shared and the latency-bound reduction levels are paid once (<= 64 sets, <= 2^27 scalars in all)
 *   gl355_bn254_g1_fixed_base_mul   out[i] = scalars[i] * base: the powers-of-tau loop of ParamsKZG::setup (verifier_api.rs:77),
 *                        8-bit windows over a per-call table, one inversion per output; out is n x 8 affine points
 *   gl355_bn254_g1_msm_prepare / _msm_prepared / _msm_bases_free   a base set that serves many MSMs (ParamsKZG's `g` / `g_lagrange`, verifier_api.rs:77:
 *                        every commitment of create_proof is an MSM over one of the two).  prepare builds once, on the device, the affine multiples
 *                        2^(c w) P_i of every base for every window w (64 B x n x windows, 6.4 GB at n = 2^23); with them the digits of ALL windows of a
 *                        scalar set fall into one set of buckets, so the bucket reduction and the per-bucket bookkeeping are paid once per set instead of
 *                        once per window, and no window sums are combined on the host.  _msm_prepared: results as gl355_bn254_g1_msm_batch over the
 *                        prepared points (scalars n_sets x n x 4).  The handle belongs to the context it was made on; free it before the context. */
int32_t gl355_bn254_fr_ntt(gl355_ctx* ctx, uint64_t* data /* n x 4 */, uint32_t log_n, int32_t inverse);
int32_t gl355_bn254_fr_coset_ntt(gl355_ctx* ctx, const uint64_t* in, uint32_t log_small, uint32_t log_n, const uint64_t shift[4], int32_t inverse,
                                 uint64_t* out);
int32_t gl355_bn254_g1_msm(gl355_ctx* ctx, const uint64_t* points /* n x 8 */, const uint64_t* scalars /* n x 4 */, uint64_t n, uint64_t result[8]);
int32_t gl355_bn254_g1_msm_batch(gl355_ctx* ctx, const uint64_t* points /* n x 8 */, const uint64_t* scalars /* n_sets x n x 4 */, uint64_t n,
                                 uint32_t n_sets, uint64_t* results /* n_sets x 8 */);
int32_t gl355_bn254_g1_fixed_base_mul(gl355_ctx* ctx, const uint64_t base[8], const uint64_t* scalars /* n x 4 */, uint64_t n, uint64_t* out /* n x 8 */);
typedef struct gl355_msm_bases gl355_msm_bases;
int32_t gl355_bn254_g1_msm_prepare(gl355_ctx* ctx, const uint64_t* points /* n x 8 */, uint64_t n, gl355_msm_bases** out);
int32_t gl355_bn254_g1_msm_prepared(gl355_ctx* ctx, const gl355_msm_bases* bases, const uint64_t* scalars /* n_sets x n x 4 */, uint32_t n_sets,
                                    uint64_t* results /* n_sets x 8 */);
int32_t gl355_bn254_g1_msm_bases_free(gl355_ctx* ctx, gl355_msm_bases* bases);

/* ---- KZG composites over those kernels: what halo2_proofs' ParamsKZG / create_proof do with them at the reference's
 * k = 23 (verifier_api.rs:77-92 `ParamsKZG::<Bn256>::setup`, `create_proof_checked`; chip/native_chip/test_utils.rs:57-95; README.md:171-177).
 * Scalars are plain 256-bit integers (4 x u64, reduced mod r on load), points affine (x | y, 8 x u64; all zero = the identity); every
 * array may be host or device memory.
 *   gl355_kzg_setup    ParamsKZG::setup with the secret handed in: g[i] = [tau^i] G1, g_lagrange[i] = [L_i(tau)] G1 (NULL to skip), G1 = (1, 2),
 *                      L_i the Lagrange basis of the 2^log_n domain.  GL355_E_INVALID_ARG if tau lies in the domain.
 *   gl355_kzg_commit   ParamsKZG::commit / commit_lagrange: sum_i poly[i] g[i].  values_form 0: poly goes with the bases as given (coefficients
 *                      with g, evaluations with g_lagrange); 1: poly are evaluations over the domain and g the MONOMIAL bases (inverse FFT, then MSM)
 *   gl355_kzg_open     eval = p(z) and witness = commit((p - p(z)) / (X - z)) for the coefficients of p: the single-point opening halo2's
 *                      multi-open provers reduce to.  quotient (optional, 2^log_n x 4) receives the quotient's coefficients.
 * A verifier with the pairing checks e(C - [eval] G1, G2) = e(witness, [tau - z] G2); with a known tau: C - [eval] G1 = [tau - z] witness. */
int32_t gl355_kzg_setup(gl355_ctx* ctx, const uint64_t tau[4], uint32_t log_n, uint64_t* g /* 2^log_n x 8 */, uint64_t* g_lagrange /* or NULL */);
int32_t gl355_kzg_commit(gl355_ctx* ctx, const uint64_t* g /* 2^log_n x 8 */, const uint64_t* poly /* 2^log_n x 4 */, uint32_t log_n, int32_t values_form,
                         uint64_t result[8]);
int32_t gl355_kzg_open(gl355_ctx* ctx, const uint64_t* g, const uint64_t* coeffs, uint32_t log_n, const uint64_t z[4], uint64_t eval[4],
                       uint64_t witness[8], uint64_t* quotient /* or NULL */);

/* ---- SURVEY 8(f) N4, the proving system itself: the data-parallel stages of halo2_proofs' create_proof::<KZGCommitmentScheme<Bn256>,
 * ProverSHPLONK<_>, _, _, Keccak256Transcript, _> as chip/native_chip/test_utils.rs:57-95 calls it (verifier_api.rs:77-92; README.md:171-177:
 * 505-511 s at k = 23).  The circuit arrives as a descriptor blob (stark-verifier_amd/halo2.py `export_desc`; the reference's chips
 * chip/native_chip/arithmetic_chip.rs:44-160 and poseidon_bn254_chip.rs:27-123 in halo2_chips.py): u64 words
 *   [0] "GL355PLK"  [1] 1  [2] k  [3] advice columns  [4] fixed columns (selectors and table columns included)  [5] instance columns
 *   [6] permutation columns  [7] lookups  [8] cs.degree()  [9] cs.blinding_factors()  [10..12] advice / fixed / instance queries
 *   [13] constants  [14] gate-program instructions  [15] gate polynomials  [16..19] the transcript's initial scalar (vk digest)  [20..23] 0
 *   then: permutation columns (kind << 32 | index; kind 0 advice, 1 fixed, 2 instance), the three query lists (column << 32 | rotation as u32),
 *   the constant pool (4 words each), the gate program, and per lookup [input instructions, table instructions] + the two programs.
 *   A program instruction is four u32 (op, dst, a, b): op 0 ADD 1 SUB 2 MUL 3 EMIT 4 NEG 5 MOV; operands kind << 24 | index with kind 0 register
 *   (< 12), 1 constant, 2 / 3 / 4 advice / fixed / instance QUERY; EMIT a = "a is the next polynomial / expression" (folded with y / theta).
 * Scalars are 4 x u64 little-endian integers, points affine x | y (8 x u64, zeros = identity), as in the KZG block above.
 *   gl355_keccak256          Keccak-256 (Ethereum's), the hash of halo2-solidity-verifier's Keccak256Transcript (test_utils.rs:73)
 *   gl355_kzg_commit_columns n_cols commitments over one resident SRS in batched MSMs: columns = [n_cols][2^log_n] scalars that go with the
 *                            bases as given (ParamsKZG::commit_lagrange per advice column, plonk/prover.rs)
 *   gl355_plonk_keygen       keygen_vk / keygen_pk (verifier_api.rs:78-79) as far as proving needs them: fixed_values [fixed columns][2^k]
 *                            scalars, mapping [permutation columns][2^k][2] u32 = the (column position, row) each cell's sigma points to
 *                            (permutation::keygen::Assembly); g / g_lagrange = ParamsKZG's two bases (device pointers are used in place and must
 *                            outlive the key, host arrays are copied).  The key lives on the context's device.  From k = 22 on it also holds the two
 *                            bases' window tables (gl355_bn254_g1_msm_prepare: 2 x 6.4 GB at k = 23) when a quarter of the free device memory covers
 *                            them; GL355_PLONK_MSM_TABLES=0 / 1 in the environment: never / from k = 12 on.  Proof bytes do not depend on it.
 *   gl355_plonk_pk_info      [k, extended k, permutation sets, quotient pieces, usable rows, proof bytes, fixed columns, permutation columns]
 *   gl355_plonk_pk_commitments   the verifying key's fixed and permutation commitments
 *   gl355_plonk_pk_digest / _set_digest   the transcript's initial scalar (halo2: vk.transcript_repr).  A descriptor whose digest field
 *                            (header words 16..19) is zero gets, at keygen, Keccak-256(descriptor words | fixed commitments | sigma
 *                            commitments, all as little-endian u64) read as a big-endian integer mod r -- the PINNED key, so circuits that
 *                            differ only in fixed values or copy constraints start different transcripts; a non-zero field is used as given;
 *                            _set_digest replaces it (a host that computed halo2's own transcript_repr)
 *   gl355_plonk_pk_export_quotient   inspection hook: the NEXT gl355_plonk_prove on this key copies the quotient's coefficient pieces h_0 .. h_{P-1}
 *                            (P = info[3]; canonical integers, [P][2^k][4] words, host memory) to host_out.  The prover builds them from degree - 1
 *                            cosets of size 2^k instead of halo2's extended domain; the hook lets a test compare the two coefficient for coefficient.
 *   gl355_plonk_prove        one proof: advice [advice columns][2^k] scalars (rows >= usable are overwritten by blinding values), instances =
 *                            the instance columns' values back to back with instance_lens[column] values each, seed = 32 bytes that fix every
 *                            random scalar (ChaCha20 block (counter = index, nonce = (stream, a, index >> 32)) -> 512-bit little-endian integer
 *                            mod r; streams 0x11 advice blinding (a = column, index = row), 0x12 permuted lookup columns (a = 2 lookup + {0
 *                            input, 1 table}), 0x13 permutation products (a = set), 0x14 lookup products (a = lookup), 0x15 the vanishing
 *                            argument's random polynomial (index = coefficient)).  trace (optional, 32 words): theta beta gamma y x and
 *                            SHPLONK's y v u.  stage_ms (optional, GL355_PLONK_STAGES doubles): wall milliseconds per stage.
 * Errors: GL355_E_INVALID_ARG for a malformed descriptor, a lookup input outside its table, or a zero grand-product denominator. */
typedef struct gl355_plonk_pk gl355_plonk_pk;
enum { GL355_PLONK_STAGE_ADVICE = 0, GL355_PLONK_STAGE_LOOKUP_PERMUTE = 1, GL355_PLONK_STAGE_PERMUTATION = 2, GL355_PLONK_STAGE_LOOKUP_PRODUCT = 3,
       GL355_PLONK_STAGE_VANISHING_RANDOM = 4, GL355_PLONK_STAGE_EVALUATE_H = 5, GL355_PLONK_STAGE_QUOTIENT_COMMIT = 6, GL355_PLONK_STAGE_EVALUATIONS = 7,
       GL355_PLONK_STAGE_SHPLONK = 8, GL355_PLONK_STAGES = 9 };
int32_t gl355_keccak256(const uint8_t* data, uint64_t len, uint8_t out[32]);
int32_t gl355_kzg_commit_columns(gl355_ctx* ctx, const uint64_t* g, const uint64_t* columns, uint32_t log_n, uint32_t n_cols, uint64_t* results /* n_cols x 8 */);
int32_t gl355_plonk_keygen(gl355_ctx* ctx, const uint64_t* desc, uint64_t desc_words, const uint64_t* g, const uint64_t* g_lagrange, const uint64_t* fixed_values,
                           const uint32_t* mapping, gl355_plonk_pk** out);
int32_t gl355_plonk_pk_info(const gl355_plonk_pk* pk, uint64_t info[8]);
int32_t gl355_plonk_pk_commitments(const gl355_plonk_pk* pk, uint64_t* fixed_commitments, uint64_t* sigma_commitments);
int32_t gl355_plonk_pk_digest(const gl355_plonk_pk* pk, uint64_t digest[4]);
int32_t gl355_plonk_pk_set_digest(gl355_plonk_pk* pk, const uint64_t digest[4]);
int32_t gl355_plonk_pk_export_quotient(gl355_plonk_pk* pk, uint64_t* host_out /* [quotient pieces][2^k][4], or NULL */);
int32_t gl355_plonk_prove(gl355_ctx* ctx, gl355_plonk_pk* pk, const uint64_t* advice, const uint64_t* instances, const uint32_t* instance_lens, const uint8_t seed[32],
                          uint8_t* proof, uint64_t capacity, uint64_t* proof_len, uint64_t* trace, double* stage_ms);
int32_t gl355_plonk_pk_destroy(gl355_plonk_pk* pk);

/* ---- a9: wires_permutation_partial_products_and_zs (vanishing_poly.rs:54-108,183-218) --------- */
int32_t gl355_zs_partial_products(gl355_ctx* ctx, const uint64_t* wires, const uint64_t* sigmas,
                                  const uint64_t* k_is, uint32_t log_n, uint32_t n_routed,
                                  uint32_t max_degree, uint64_t beta, uint64_t gamma,
                                  uint64_t* z_out, uint64_t* pp_out);

#ifdef __cplusplus
}
#endif
#endif /* GL355_H */
