// CircuitData::prove for B independent units of ONE circuit in lock-step on one prover context
// (src/plonky2_semaphore/access_set.rs:94, recursion.rs:168, wrapper.rs:55; the reference proves its units from a rayon
// par_iter, recursion.rs:214-227,300-308).
//
// Why lock-step batches: a single proof of n = 2^13..2^14 is ~430 kernel launches of which ~350 are shorter than 30 us (Merkle
// levels of a few thousand nodes, FRI layers, scans) and ~20 host round trips for the Fiat-Shamir transcript -- the device idles
// between them and the small levels run the 16-lanes-per-node permutation (3x the instructions) only to cut latency.  Giving
// every stage a unit dimension makes each launch B times larger (Merkle levels reach the one-lane-per-node kernel B levels
// earlier, small kernels fill the chip), divides launches and host round trips per unit by B, and keeps every unit's bytes
// exactly what a one-by-one proof produces: the units never interact, each has its own transcript (host, ~50 permutations),
// challenges, blinding key and proof buffer.  B = 1 is the single-proof entry (gl355_prove, gl355_prove_sparse).
//
// Stage order and transcript: chip/plonk/plonk_verifier_chip.rs:55-154; oracle and opening order: types/common_data.rs:100-222,
// types/assigned.rs:26-44; permutation argument: chip/plonk/vanishing_poly.rs:54-108,183-218; DEEP / FRI:
// chip/fri_chip.rs:112-149,168-226,275-327,364-376.
//
// Layout in HBM (all unit-major, uniform strides): coefficients [B * batch][n], LDE [B * batch][N] column-major in bit-reversed
// row order (what the DIF transform writes), salt columns [B * 4][N] apart from the LDE (so the NTT sees one uniform batch of
// B * batch columns; the leaf hash and the query openings read the salt segment for the last four leaf elements), digests
// [B][..] in plonky2's layout, caps [B][2^cap_height][4].  B trees of 2^cap_height cap subtrees are, to the Merkle kernels,
// one forest of B * 2^cap_height subtrees.
#include "gl355_internal.h"
#include "blinding.cuh"
#include "poseidon.cuh"

using namespace gl355;

namespace gl355 {

constexpr uint32_t MAXB = GL355_MAX_UNITS;

// ---- committed batches of B units ------------------------------------------------------------------------------------------
struct BOracle {
    Ctx* ctx = nullptr;
    uint32_t B = 0, log_n = 0, rate_bits = 0, batch = 0, leaf_len = 0, cap_height = 0;
    uint64_t* base = nullptr;
    uint64_t *coeffs = nullptr, *lde = nullptr, *salt = nullptr, *digests = nullptr, *cap = nullptr;
    uint64_t n_dig = 0;      // digests per unit
    ~BOracle() { if (base && ctx) ctx->release(base); }
    int32_t alloc(Ctx* c, uint32_t b, uint32_t lg, uint32_t rb, uint32_t bat, bool salted, uint32_t cap_h) {
        ctx = c; B = b; log_n = lg; rate_bits = rb; batch = bat; leaf_len = bat + (salted ? GL355_SALT_SIZE : 0); cap_height = cap_h;
        const uint64_t n = 1ull << lg, N = n << rb, n_cap = 1ull << cap_h;
        n_dig = 2 * (N - n_cap);
        const uint64_t total = (uint64_t)B * ((uint64_t)bat * n + (uint64_t)leaf_len * N + n_dig * 4 + n_cap * 4);
        void* p = nullptr;
        GL355_TRY(c->alloc(total * 8, &p));
        base = reinterpret_cast<uint64_t*>(p);
        coeffs = base;
        lde = coeffs + (uint64_t)B * bat * n;
        salt = salted ? lde + (uint64_t)B * bat * N : nullptr;
        digests = lde + (uint64_t)B * leaf_len * N;
        cap = digests + (uint64_t)B * n_dig * 4;
        return GL355_OK;
    }
};

// one oracle as the kernels see it: unit u's data at base + u * stride (stride 0 = shared by all units: constants_sigmas)
struct OView {
    const uint64_t* coeffs; uint64_t coeffs_us;
    const uint64_t* lde; uint64_t lde_us;
    const uint64_t* salt; uint64_t salt_us;
    const uint64_t* digests; uint64_t dig_us;
    uint32_t batch, leaf_len;
};
static OView view_of(const BOracle& o) {
    const uint64_t n = 1ull << o.log_n, N = n << o.rate_bits;
    return OView{o.coeffs, (uint64_t)o.batch * n, o.lde, (uint64_t)o.batch * N, o.salt, (uint64_t)GL355_SALT_SIZE * N, o.digests, o.n_dig * 4, o.batch, o.leaf_len};
}
static OView view_of(const gl355_oracle* o) {
    return OView{o->coeffs, 0, o->lde, 0, nullptr, 0, o->digests, 0, o->batch, o->leaf_len};
}

struct UnitKeys { BlindKey k[MAXB]; };
struct UnitVals { uint64_t v[MAXB * 4]; };      // up to 4 words per unit (challenges, pi hash, extension elements)

// ---- witness ---------------------------------------------------------------------------------------------------------------
// scatter the sparse witness rows of every unit into its column-major wire matrix (rows: [B][n_rows][num_wires])
__global__ void witness_rows_units_kernel(uint64_t* wires, uint64_t n, uint32_t num_wires, const uint32_t* row_idx, const uint64_t* row_vals,
                                          uint32_t n_rows) {
    const uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t per = (uint64_t)n_rows * num_wires;
    if (g >= per) return;
    const uint32_t u = blockIdx.y, r = g / num_wires, c = g % num_wires;
    wires[((uint64_t)u * num_wires + c) * n + row_idx[r]] = gl_canon(row_vals[(uint64_t)u * per + g]);
}
// element g of unit u's witness-blinding stream: g < n_blind * num_wires fills wire g / n_blind of blinding row g % n_blind; the next
// n_z_pairs * num_routed elements are the Z-blinding pairs: element c * n_z_pairs + k is the value of routed column c on BOTH rows of
// pair k (plonky2 `blind`: every routed wire of a pair is random and copy-constrained between its two rows -- one value per pair on
// column 0 only, as in round 2, left Z_0 and Z_1 at a blinding row functions of the same value)
__global__ void witness_blind_units_kernel(uint64_t* wires, uint64_t n, uint32_t num_wires, uint32_t num_routed, uint32_t blind_start, uint32_t n_blind,
                                           uint32_t z_start, uint32_t n_z_pairs, UnitKeys keys) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t n_a = (uint64_t)n_blind * num_wires, n_z = (uint64_t)n_z_pairs * num_routed;
    if (4 * b >= n_a + n_z) return;
    const uint32_t u = blockIdx.y;
    uint64_t* w = wires + (uint64_t)u * num_wires * n;
    uint64_t e[4];
    blind_block_elements(keys.k[u], GL355_BLIND_STREAM_WITNESS, (uint32_t)b, e);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t g = 4 * b + j;
        if (g < n_a) {
            const uint32_t c = g / n_blind, r = g % n_blind;
            w[(uint64_t)c * n + blind_start + r] = e[j];
        } else if (g < n_a + n_z) {
            const uint64_t h = g - n_a, c = h / n_z_pairs, k = h % n_z_pairs;
            w[c * n + z_start + 2 * k] = e[j];
            w[c * n + z_start + 2 * k + 1] = e[j];
        }
    }
}
// salt columns of a blinded oracle, written straight into leaf (bit-reversed row) order: stream element c * N + i is the salt of
// column c at natural row i (what plonky2 appends to the LDE before its transpose), stored at row bitrev(i)
__global__ void salt_units_kernel(uint64_t* salt, uint32_t lde_bits, UnitKeys keys, uint32_t stream) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t N = 1ull << lde_bits, cnt = (uint64_t)GL355_SALT_SIZE * N;
    if (4 * b >= cnt) return;
    const uint32_t u = blockIdx.y;
    uint64_t e[4];
    blind_block_elements(keys.k[u], stream, (uint32_t)b, e);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t g = 4 * b + j;          // N >= 4 is a multiple of 4: the four elements are in one column
        const uint64_t c = g >> lde_bits, i = g & (N - 1);
        const uint64_t r = __brevll(i) >> (64 - lde_bits);
        salt[((uint64_t)u * GL355_SALT_SIZE + c) * N + r] = e[j];
    }
}

// ---- a9: permutation argument, instance = (unit, challenge) ---------------------------------------------------------------------
struct ZsArgs {
    const uint64_t* wires; uint64_t wires_us;     // [B][num_wires][n]
    const uint64_t* sigmas; const uint64_t* k_is;
    uint32_t log_n, n_routed, max_degree, nch, n_chunks, npp;
    uint64_t g;
    uint64_t* chunk_q;    // [inst][n_chunks][n]
    uint64_t* row_prod;   // [inst][n]
    uint64_t* zbuf; uint64_t z_us;                // unit u: [Z_0..Z_{nch-1} | pp_{0,*} | pp_{1,*} ...][n]
    UnitVals betas, gammas;                       // [u * 4 + k]
};
constexpr int ZS_MAX_CHUNKS = 16;
// per row: the n_chunks chunk quotients prod(num)/prod(den) and their product; ONE field inversion per row (prefix products of
// the chunk denominators, inverse of the total, walked back) instead of one per chunk
__global__ void __launch_bounds__(256) zs_rows_units_kernel(ZsArgs a) {
    const uint64_t n = 1ull << a.log_n;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t inst = blockIdx.y, u = inst / a.nch, k = inst % a.nch;
    const uint64_t beta = a.betas.v[u * 4 + k], gamma = a.gammas.v[u * 4 + k];
    const uint64_t* wires = a.wires + (uint64_t)u * a.wires_us;
    const uint64_t x = gl_pow(a.g, i);
    const uint64_t bx = gl_mul(beta, x);
    uint64_t num[ZS_MAX_CHUNKS], pre[ZS_MAX_CHUNKS], den[ZS_MAX_CHUNKS];
    uint64_t run = 1;
#pragma unroll
    for (int ch = 0; ch < ZS_MAX_CHUNKS; ch++) {
        num[ch] = 1; den[ch] = 1; pre[ch] = run;
        if ((uint32_t)ch < a.n_chunks) {
            uint64_t nm = 1, dn = 1;
            for (uint32_t j = ch * a.max_degree; j < (ch + 1) * a.max_degree && j < a.n_routed; j++) {
                const uint64_t w = wires[(uint64_t)j * n + i];
                nm = gl_mul(nm, gl_add(gl_add(w, gl_mul(bx, a.k_is[j])), gamma));
                dn = gl_mul(dn, gl_add(gl_add(w, gl_mul(beta, a.sigmas[(uint64_t)j * n + i])), gamma));
            }
            num[ch] = nm; den[ch] = dn;
            run = gl_mul(run, dn);
        }
    }
    uint64_t inv = gl_inv(run);          // 1 / prod_ch den[ch]
    uint64_t rp = 1;
    uint64_t* cq = a.chunk_q + (uint64_t)inst * a.n_chunks * n;
#pragma unroll
    for (int ch = ZS_MAX_CHUNKS - 1; ch >= 0; ch--) {
        if ((uint32_t)ch < a.n_chunks) {
            const uint64_t dinv = gl_mul(inv, pre[ch]);      // 1 / den[ch]
            inv = gl_mul(inv, den[ch]);
            const uint64_t q = gl_mul(num[ch], dinv);
            cq[(uint64_t)ch * n + i] = q;
            rp = gl_mul(rp, q);
        }
    }
    a.row_prod[(uint64_t)inst * n + i] = rp;
}
// one workgroup per instance: z[i] = prod_{j<i} row_prod[j], tile-by-tile log-step product scan
__global__ void __launch_bounds__(1024) zs_scan_units_kernel(ZsArgs a) {
    __shared__ uint64_t sh[1024];
    const int tid = threadIdx.x;
    const uint32_t inst = blockIdx.x, u = inst / a.nch, k = inst % a.nch;
    const uint64_t n = 1ull << a.log_n;
    const uint64_t* row_prod = a.row_prod + (uint64_t)inst * n;
    uint64_t* z = a.zbuf + (uint64_t)u * a.z_us + (uint64_t)k * n;
    uint64_t running = 1;
    for (uint64_t base = 0; base < n; base += 1024) {
        const uint64_t i = base + tid;
        uint64_t v = i < n ? row_prod[i] : 1;
        sh[tid] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            uint64_t o = tid >= s ? sh[tid - s] : 1;
            __syncthreads();
            v = gl_mul(v, o);
            sh[tid] = v;
            __syncthreads();
        }
        const uint64_t excl = tid ? sh[tid - 1] : 1;
        if (i < n) z[i] = gl_canon(gl_mul(running, excl));
        const uint64_t tile_total = sh[1023];
        __syncthreads();
        running = gl_mul(running, tile_total);
    }
}
// partial products: acc = z[i]; acc *= q[ch][i]; pp[ch][i] = acc  (ch < n_chunks - 1)
__global__ void zs_partials_units_kernel(ZsArgs a) {
    const uint64_t n = 1ull << a.log_n;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t inst = blockIdx.y, u = inst / a.nch, k = inst % a.nch;
    uint64_t* zu = a.zbuf + (uint64_t)u * a.z_us;
    const uint64_t* cq = a.chunk_q + (uint64_t)inst * a.n_chunks * n;
    uint64_t* pp = zu + ((uint64_t)a.nch + (uint64_t)k * a.npp) * n;
    uint64_t acc = zu[(uint64_t)k * n + i];
    for (uint32_t ch = 0; ch + 1 < a.n_chunks; ch++) {
        acc = gl_mul(acc, cq[(uint64_t)ch * n + i]);
        pp[(uint64_t)ch * n + i] = gl_canon(acc);
    }
}

// ---- openings and DEEP quotient ----------------------------------------------------------------------------------------------
// the polynomials of an opening batch, by index, without a pointer table: oracle o contributes count[o] columns of n coefficients
struct PolySet {
    const uint64_t* base[4]; uint64_t us[4]; uint32_t count[4];
    uint32_t n_sets, log_n;
};
GL_DEV const uint64_t* poly_ptr(const PolySet& s, uint32_t u, uint32_t i) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if ((uint32_t)k + 1 < s.n_sets && i >= s.count[o]) { i -= s.count[o]; o++; }
    return s.base[o] + (uint64_t)u * s.us[o] + ((uint64_t)i << s.log_n);
}

// OpeningSet::new: block (i, u) evaluates polynomial i of unit u at zeta_u (i < n_all) or Z polynomial i - n_all at g * zeta_u;
// lane t Horner-evaluates the coefficients k = t (mod 256) in z^256 (coalesced loads) and scales by z^t; the partials are summed in LDS.  out[u][i] (extension)
struct EvalArgs { PolySet all, zs; uint32_t n_all; UnitVals zeta /* [u*4+0..1] = zeta, [u*4+2..3] = g * zeta */; uint64_t* out; uint32_t out_us; };
__global__ void __launch_bounds__(256) eval_polys_units_kernel(EvalArgs a) {
    __shared__ uint64_t sh[512];
    const int tid = threadIdx.x;
    const uint32_t u = blockIdx.y, i = blockIdx.x;
    const bool at_next = i >= a.n_all;
    const uint64_t* p = at_next ? poly_ptr(a.zs, u, i - a.n_all) : poly_ptr(a.all, u, i);
    const gl2 zz = at_next ? gl2_make(a.zeta.v[u * 4 + 2], a.zeta.v[u * 4 + 3]) : gl2_make(a.zeta.v[u * 4], a.zeta.v[u * 4 + 1]);
    const uint64_t n = 1ull << a.all.log_n;
    const gl2 acc = gl2_horner_strided256(p, n, zz, (uint32_t)tid);
    sh[2 * tid] = acc.c0; sh[2 * tid + 1] = acc.c1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            sh[2 * tid] = gl_add(sh[2 * tid], sh[2 * (tid + s)]);
            sh[2 * tid + 1] = gl_add(sh[2 * tid + 1], sh[2 * (tid + s) + 1]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        uint64_t* o = a.out + (uint64_t)u * a.out_us + 2ull * i;
        o[0] = gl_canon(sh[0]); o[1] = gl_canon(sh[1]);
    }
}

constexpr int DEEP_BLK = 256;  // coefficients per workgroup in the division scan
// per-unit tables of the two opening batches (point A = zeta, point B = g * zeta), ext elements:
//   [0, n_alpha]            alpha^i, i <= n_alpha (n_alpha = number of polynomials of the larger batch)
//   then for A and for B:   z^(2^s), s < 8 | z^e, e <= DEEP_BLK
GL_HD uint32_t deep_tab_len(uint32_t n_alpha) { return (n_alpha + 1) + 2 * (8 + DEEP_BLK + 1); }
struct DeepTabArgs { uint64_t* tab; uint32_t n_alpha; UnitVals alpha /* [u*4+0..1] */, zeta /* as EvalArgs */; };
__global__ void __launch_bounds__(256) deep_tables_kernel(DeepTabArgs a) {
    const uint32_t u = blockIdx.x;
    uint64_t* t = a.tab + (uint64_t)u * deep_tab_len(a.n_alpha) * 2;
    const gl2 al = gl2_make(a.alpha.v[u * 4], a.alpha.v[u * 4 + 1]);
    for (uint32_t i = threadIdx.x; i <= a.n_alpha; i += blockDim.x) {
        const gl2 c = gl2_canon(gl2_pow(al, i));
        t[2 * i] = c.c0; t[2 * i + 1] = c.c1;
    }
    for (int pt = 0; pt < 2; pt++) {
        const gl2 z = gl2_make(a.zeta.v[u * 4 + 2 * pt], a.zeta.v[u * 4 + 2 * pt + 1]);
        uint64_t* tz = t + 2 * ((uint64_t)(a.n_alpha + 1) + (uint64_t)pt * (8 + DEEP_BLK + 1));
        for (uint32_t i = threadIdx.x; i < 8 + DEEP_BLK + 1; i += blockDim.x) {
            const gl2 c = gl2_canon(i < 8 ? gl2_pow(z, 1ull << i) : gl2_pow(z, i - 8));
            tz[2 * i] = c.c0; tz[2 * i + 1] = c.c1;
        }
    }
}
struct DeepArgs {
    PolySet polys; uint32_t n_polys, n_alpha, point;   // point 0 = zeta tables, 1 = g * zeta tables
    const uint64_t* tab;
    uint64_t n, n_blocks;
    uint64_t* v;        // [B][n] ext scratch
    uint64_t* totals;   // [B][n_blocks] ext
    uint64_t* carry;    // [B][n_blocks] ext
    uint64_t* acc;      // [B][2][n]: the accumulated quotient as two base-field columns (what the F_p^2 LDE takes)
};
GL_DEV const uint64_t* deep_tab_u(const DeepArgs& a, uint32_t u) { return a.tab + (uint64_t)u * deep_tab_len(a.n_alpha) * 2; }
// phase 1: comp[k] = sum_i alpha^i p_i[k]; in-block suffix Horner sums v[k] = sum_{j>=k, j in block} comp[j] z^(j-k)
__global__ void __launch_bounds__(DEEP_BLK) deep_reduce_scan_units_kernel(DeepArgs a) {
    __shared__ uint64_t sh[2 * DEEP_BLK];
    const int tid = threadIdx.x;
    const uint32_t u = blockIdx.y;
    const uint64_t k = blockIdx.x * (uint64_t)DEEP_BLK + tid;
    const uint64_t* tab = deep_tab_u(a, u);
    const uint64_t* z_pow2 = tab + 2 * ((uint64_t)(a.n_alpha + 1) + (uint64_t)a.point * (8 + DEEP_BLK + 1));
    gl2 c = gl2_make(0, 0);
    if (k < a.n) {
        for (uint32_t i = 0; i < a.n_polys; i++) {
            const uint64_t coef = poly_ptr(a.polys, u, i)[k];
            c.c0 = gl_add(c.c0, gl_mul(tab[2 * i], coef));
            c.c1 = gl_add(c.c1, gl_mul(tab[2 * i + 1], coef));
        }
    }
    sh[2 * tid] = c.c0; sh[2 * tid + 1] = c.c1;
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 8; s++) {
        const int other = tid + (1 << s);
        gl2 add = gl2_make(0, 0);
        if (other < DEEP_BLK) add = gl2_mul(gl2_make(sh[2 * other], sh[2 * other + 1]), gl2_make(z_pow2[2 * s], z_pow2[2 * s + 1]));
        __syncthreads();
        c = gl2_add(c, add);
        sh[2 * tid] = c.c0; sh[2 * tid + 1] = c.c1;
        __syncthreads();
    }
    uint64_t* v = a.v + (uint64_t)u * a.n * 2;
    if (k < a.n) { v[2 * k] = c.c0; v[2 * k + 1] = c.c1; }
    if (tid == 0) { uint64_t* t = a.totals + ((uint64_t)u * a.n_blocks + blockIdx.x) * 2; t[0] = c.c0; t[1] = c.c1; }
}
// phase 2 (one lane per unit): carry[blk] = T_{blk+1} + z^B carry[blk+1]
__global__ void deep_carry_units_kernel(DeepArgs a) {
    const uint32_t u = blockIdx.x;
    if (threadIdx.x != 0) return;
    const uint64_t* tab = deep_tab_u(a, u);
    const uint64_t* z_pows = tab + 2 * ((uint64_t)(a.n_alpha + 1) + (uint64_t)a.point * (8 + DEEP_BLK + 1) + 8);
    const gl2 zb = gl2_make(z_pows[2 * DEEP_BLK], z_pows[2 * DEEP_BLK + 1]);
    const uint64_t* totals = a.totals + (uint64_t)u * a.n_blocks * 2;
    uint64_t* carry = a.carry + (uint64_t)u * a.n_blocks * 2;
    gl2 c = gl2_make(0, 0);
    for (uint64_t blk = a.n_blocks; blk-- > 0;) {
        carry[2 * blk] = c.c0; carry[2 * blk + 1] = c.c1;
        c = gl2_add(gl2_make(totals[2 * blk], totals[2 * blk + 1]), gl2_mul(zb, c));
    }
}
// phase 3: quotient q_k = b_{k+1}, b_j = v[j] + z^(B - j%B) carry[blk(j)]; acc = acc * alpha^n_polys + q
__global__ void __launch_bounds__(DEEP_BLK) deep_finish_units_kernel(DeepArgs a) {
    const uint64_t k = blockIdx.x * (uint64_t)DEEP_BLK + threadIdx.x;
    if (k >= a.n) return;
    const uint32_t u = blockIdx.y;
    const uint64_t* tab = deep_tab_u(a, u);
    const uint64_t* z_pows = tab + 2 * ((uint64_t)(a.n_alpha + 1) + (uint64_t)a.point * (8 + DEEP_BLK + 1) + 8);
    const uint64_t* v = a.v + (uint64_t)u * a.n * 2;
    const uint64_t* carry = a.carry + (uint64_t)u * a.n_blocks * 2;
    gl2 q = gl2_make(0, 0);
    const uint64_t j = k + 1;
    if (j < a.n) {
        const uint64_t blk = j / DEEP_BLK, off = j % DEEP_BLK;
        const gl2 cr = gl2_make(carry[2 * blk], carry[2 * blk + 1]);
        const uint64_t e = DEEP_BLK - off;
        q = gl2_add(gl2_make(v[2 * j], v[2 * j + 1]), gl2_mul(gl2_make(z_pows[2 * e], z_pows[2 * e + 1]), cr));
    }
    uint64_t* acc0 = a.acc + (uint64_t)u * 2 * a.n;
    uint64_t* acc1 = acc0 + a.n;
    const gl2 shift = gl2_make(tab[2 * a.n_polys], tab[2 * a.n_polys + 1]);
    const gl2 r = gl2_canon(gl2_add(gl2_mul(gl2_make(acc0[k], acc1[k]), shift), q));
    acc0[k] = r.c0; acc1[k] = r.c1;
}

// ---- a12: FRI commit phase -----------------------------------------------------------------------------------------------------
// layer leaves from the bit-reversed F_p^2 evaluations held as two base columns per unit: leaf i = ext values 2i, 2i+1 of the
// bit-reversed sequence (fri_chip.rs:275-316) = (c0[2i], c1[2i], c0[2i+1], c1[2i+1])
__global__ void fri_leaves_units_kernel(const uint64_t* cols /* [B][2][len] */, uint64_t len, uint64_t* leaves /* [B][len/2][4] */) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= len / 2) return;
    const uint32_t u = blockIdx.y;
    const uint64_t* c0 = cols + (uint64_t)u * 2 * len;
    const uint64_t* c1 = c0 + len;
    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(c0 + 2 * i);
    const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(c1 + 2 * i);
    uint64_t* dst = leaves + ((uint64_t)u * (len / 2) + i) * 4;
    *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(a.x, b.x);
    *reinterpret_cast<ulonglong2*>(dst + 2) = make_ulonglong2(a.y, b.y);
}
// arity-2 fold of the coefficient columns: out[k] = c[2k] + beta_u * c[2k+1]
__global__ void fri_fold_units_kernel(const uint64_t* cols /* [B][2][len] */, uint64_t len, UnitVals beta, uint64_t* out /* [B][2][len/2] */) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t half = len / 2;
    if (k >= half) return;
    const uint32_t u = blockIdx.y;
    const uint64_t* c0 = cols + (uint64_t)u * 2 * len;
    const uint64_t* c1 = c0 + len;
    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(c0 + 2 * k);
    const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(c1 + 2 * k);
    const gl2 r = gl2_canon(gl2_add(gl2_make(a.x, b.x), gl2_mul(gl2_make(a.y, b.y), gl2_make(beta.v[u * 4], beta.v[u * 4 + 1]))));
    uint64_t* o0 = out + (uint64_t)u * 2 * half;
    o0[k] = r.c0; o0[half + k] = r.c1;
}
// proof of work of every unit that still lacks a witness: candidates start + g, the smallest passing one of the launch wins
struct PowArgs { uint64_t state[MAXB * 12]; uint32_t pos[MAXB]; uint32_t todo[MAXB]; uint32_t bits; uint64_t start; unsigned long long* best; };
__global__ void __launch_bounds__(256) pow_grind_units_kernel(PowArgs a) {
    const uint32_t u = blockIdx.y;
    if (!a.todo[u]) return;
    const uint64_t w = a.start + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    // the smallest witness wins, so a candidate above the best one found so far cannot matter: workgroups are dispatched in
    // roughly increasing order, and once a witness is known the rest of the launch exits here (expected work ~2^bits instead of the
    // 2^(bits+1) candidates of the launch); candidates below the current best are never skipped, so the result is still the minimum
    if (w > __hip_atomic_load(a.best + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = a.state[u * 12 + k];
#pragma unroll
    for (int k = 0; k < 12; k++) if ((uint32_t)k == a.pos[u]) s[k] = w;
    psd_permute(s);
    const uint64_t resp = gl_canon(s[7]);
    if (a.bits == 0 || (resp >> (64 - a.bits)) == 0) atomicMin(a.best + u, (unsigned long long)w);
}

// ---- a14: query openings ---------------------------------------------------------------------------------------------------------
// block (q, u): row idx[u][q] of unit u's column-major LDE (+ salt segment) and its Merkle path, dense per-unit outputs
struct OpenArgs { OView o; uint64_t N; uint32_t lde_bits, cap_height, n_idx; const uint64_t* idx /* [B][n_idx] */; uint64_t* leaves; uint64_t* sibs; };
__global__ void open_units_kernel(OpenArgs a) {
    const uint32_t q = blockIdx.x, u = blockIdx.y;
    const uint64_t index = a.idx[(uint64_t)u * a.n_idx + q];
    const uint32_t layers = a.lde_bits - a.cap_height;
    const uint64_t* lde = a.o.lde + (uint64_t)u * a.o.lde_us;
    const uint64_t* salt = a.o.salt ? a.o.salt + (uint64_t)u * a.o.salt_us : nullptr;
    uint64_t* lo = a.leaves + ((uint64_t)u * a.n_idx + q) * a.o.leaf_len;
    for (uint32_t c = threadIdx.x; c < a.o.leaf_len; c += blockDim.x)
        lo[c] = c < a.o.batch ? lde[(uint64_t)c * a.N + index] : salt[(uint64_t)(c - a.o.batch) * a.N + index];
    const uint64_t sub_leaves = 1ull << layers;
    const uint64_t* tree = a.o.digests + (uint64_t)u * a.o.dig_us + (index >> layers) * 2 * (sub_leaves - 1) * 4;
    const uint64_t k0 = index & (sub_leaves - 1);
    uint64_t* so = a.sibs + ((uint64_t)u * a.n_idx + q) * layers * 4;
    for (uint32_t e = threadIdx.x; e < layers * 4; e += blockDim.x) {
        const uint32_t layer = e >> 2;
        const uint64_t k = (k0 >> layer) ^ 1;
        so[e] = tree[digest_slot(layer, k) * 4 + (e & 3)];
    }
}
// block (q, u, l): the pair of layer l at index idx[u][q] >> (l + 1) and its path
struct OpenFriArgs {
    const uint64_t* trees; uint64_t leaf_off[32], dig_off[32];     // layer l: leaves [B][nl][4] at leaf_off, digests [B][2(nl - cap)][4] at dig_off
    uint64_t sib_off[32], sib_total;
    uint32_t lde_bits, cap_height, n_idx, n_layers;
    const uint64_t* idx; uint64_t* evals /* [B][n_idx][n_layers][4] */; uint64_t* sibs /* [B][n_idx][sib_total] */;
};
__global__ void open_fri_units_kernel(OpenFriArgs a) {
    const uint32_t q = blockIdx.x, u = blockIdx.y, l = blockIdx.z;
    const uint64_t index = a.idx[(uint64_t)u * a.n_idx + q] >> (l + 1);
    const uint32_t log_nl = a.lde_bits - 1 - l, layers = log_nl - a.cap_height;
    const uint64_t nl = 1ull << log_nl, n_cap = 1ull << a.cap_height;
    const uint64_t* leaves = a.trees + a.leaf_off[l] + (uint64_t)u * nl * 4;
    const uint64_t* digs = a.trees + a.dig_off[l] + (uint64_t)u * 2 * (nl - n_cap) * 4;
    uint64_t* ev = a.evals + (((uint64_t)u * a.n_idx + q) * a.n_layers + l) * 4;
    if (threadIdx.x < 4) ev[threadIdx.x] = leaves[index * 4 + threadIdx.x];
    const uint64_t sub_leaves = 1ull << layers;
    const uint64_t* tree = digs + (index >> layers) * 2 * (sub_leaves - 1) * 4;
    const uint64_t k0 = index & (sub_leaves - 1);
    uint64_t* so = a.sibs + ((uint64_t)u * a.n_idx + q) * a.sib_total + a.sib_off[l];
    for (uint32_t e = threadIdx.x; e < layers * 4; e += blockDim.x) {
        const uint32_t layer = e >> 2;
        const uint64_t k = (k0 >> layer) ^ 1;
        so[e] = tree[digest_slot(layer, k) * 4 + (e & 3)];
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK(ctx) GL355_HIP(ctx, hipGetLastError())

// PolynomialBatch::from_values / from_coeffs for B units: o.coeffs already holds the values (or coefficients) [B * batch][n]
static int32_t commit_units(Ctx* ctx, int32_t hasher, BOracle& o, bool is_coeffs, bool canonical, const UnitKeys* keys, uint32_t stream_id) {
    const uint64_t n = 1ull << o.log_n, N = n << o.rate_bits;
    const uint32_t cols = o.B * o.batch, lde_bits = o.log_n + o.rate_bits;
    if (!is_coeffs) GL355_TRY(ntt_dev(ctx, o.coeffs, o.log_n, cols, n, true, 0));
    else if (!canonical) GL355_TRY(canon_dev(ctx, o.coeffs, (uint64_t)cols * n));
    GL355_TRY(lde_dev(ctx, o.coeffs, n, o.log_n, o.rate_bits, GL355_COSET_SHIFT, cols, o.lde, N, true));
    if (o.salt) {
        const uint64_t cnt = (uint64_t)GL355_SALT_SIZE * N;
        ProfScope ps(ctx, "salt", cnt * 8 * o.B);
        hipLaunchKernelGGL(salt_units_kernel, dim3((uint32_t)((cnt / 4 + 255) / 256), o.B), dim3(256), 0, ctx->stream, o.salt, lde_bits, *keys, stream_id);
        LAUNCH_CHECK(ctx);
    }
    LeafArgs a;
    memset(&a, 0, sizeof a);
    a.leaves = o.lde; a.n_leaves = (uint64_t)o.B * N; a.leaf_len = o.leaf_len; a.col_major = 1; a.stride = N;
    a.unit_log = lde_bits; a.n_main = o.batch; a.unit_stride = (uint64_t)o.batch * N; a.salt = o.salt; a.salt_unit_stride = (uint64_t)GL355_SALT_SIZE * N;
    return merkle_build_args_any(ctx, hasher, a, lde_bits - o.cap_height, o.digests, o.cap);
}

int32_t prove_units(Ctx* ctx, const gl355_prover_data* pd, uint32_t B, const uint64_t* d_wires_dense, const uint32_t* row_idx,
                    const uint64_t* rows_host, uint32_t n_rows, uint32_t blind_start, uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs,
                    const ProveUnit* io) {
    if (!pd || !pd->circuit || !pd->constants_sigmas || !pd->sigmas || !pd->k_is || !io || B == 0 || B > MAXB)
        return ctx->fail(GL355_E_INVALID_ARG, "prove: null argument or more than GL355_MAX_UNITS units");
    struct CpuScope {      // thread CPU time of the whole call, for gl355_profile_read's host:cpu_in_prove
        Ctx* c; uint64_t t0;
        explicit CpuScope(Ctx* cx) : c(cx), t0(cx->prof_on ? thread_cpu_ns() : 0) {}
        ~CpuScope() { if (c->prof_on) { c->prove_cpu_ns += thread_cpu_ns() - t0; c->prove_calls++; } }
    } cpu_scope(ctx);
    const gl355_circuit& c = *pd->circuit;
    const uint32_t nch = c.num_challenges, qdf = c.max_degree, npp = c.num_partial_products, routed = c.num_routed_wires, nw = c.num_wires;
    const uint32_t lde_bits = c.degree_bits + c.rate_bits, cap_h = pd->cap_height, L = pd->n_fri_layers, nq_idx = pd->num_queries;
    const uint64_t n = 1ull << c.degree_bits, N = 1ull << lde_bits, n_cap = 1ull << cap_h;
    if (nch == 0 || nch > 4 || L > 32 || lde_bits > 27) return ctx->fail(GL355_E_UNSUPPORTED, "prove: unsupported shape");
    if (L >= c.degree_bits) return ctx->fail(GL355_E_UNSUPPORTED, "prove: the final polynomial must keep at least two coefficients");
    if (lde_bits - L < cap_h + 1u) return ctx->fail(GL355_E_INVALID_ARG, "prove: too many FRI layers for the cap height");
    const uint32_t n_chunks = (routed + qdf - 1) / qdf;
    if (n_chunks > ZS_MAX_CHUNKS || npp + 1 != n_chunks) return ctx->fail(GL355_E_UNSUPPORTED, "prove: at most 16 partial-product chunks");
    const gl355_oracle* cs = pd->constants_sigmas;
    if (cs->log_n != c.degree_bits || cs->rate_bits != c.rate_bits || cs->cap_height != cap_h)
        return ctx->fail(GL355_E_INVALID_ARG, "prove: constants_sigmas oracle does not match the circuit");
    const int32_t hasher = pd->hasher;
    if (hasher != GL355_HASH_POSEIDON && hasher != GL355_HASH_BN254_POSEIDON) return ctx->fail(GL355_E_INVALID_ARG, "prove: unknown hasher");
    if (cs->hasher != hasher) return ctx->fail(GL355_E_INVALID_ARG, "prove: constants_sigmas was committed with another hasher");
    const bool zk = pd->zero_knowledge != 0;
    const uint64_t need = gl355_proof_words(pd);
    for (uint32_t u = 0; u < B; u++) {
        if (!io[u].proof || (!io[u].public_inputs && io[u].n_public_inputs)) return ctx->fail(GL355_E_INVALID_ARG, "prove: null argument");
        if (io[u].proof_capacity_words < need) return ctx->fail(GL355_E_INVALID_ARG, "prove: proof buffer too small (see gl355_proof_words)");
    }
    uint32_t qdb = 0;
    while ((1u << qdb) < qdf) qdb++;
    const uint64_t nq = n << qdb;

    // ---- transcripts, one per unit (host) ---------------------------------------------------------------------------------
    std::vector<gl355_challenger> ch(B);
    std::vector<uint64_t*> out(B);
    UnitKeys keys;
    UnitVals pi_hashes;
    memset(&pi_hashes, 0, sizeof pi_hashes);
    for (uint32_t u = 0; u < B; u++) {
        uint64_t* hdr = io[u].proof;
        hdr[0] = need; hdr[1] = c.degree_bits; hdr[2] = L; hdr[3] = nq_idx; hdr[4] = io[u].n_public_inputs; hdr[5] = zk; hdr[6] = cap_h; hdr[7] = nch;
        out[u] = hdr + 8;
        gl355_challenger_init_h(&ch[u], hasher);
        gl355_host_hash_no_pad(io[u].public_inputs, io[u].n_public_inputs, &pi_hashes.v[u * 4]);
        gl355_challenger_observe(&ch[u], pd->circuit_digest, 4);
        gl355_challenger_observe(&ch[u], &pi_hashes.v[u * 4], 4);
        memcpy(keys.k[u].w, io[u].key, 32);
    }
    // pinned staging for everything that crosses PCIe (caps, openings, final polynomials, query openings)
    const uint64_t n_open_all = (uint64_t)cs->batch + nw + (uint64_t)nch * (1 + npp) + (uint64_t)nch * qdf;
    const uint32_t depth0 = lde_bits - cap_h;
    uint64_t sib_total = 0;
    for (uint32_t l = 0; l < L; l++) sib_total += (uint64_t)(lde_bits - 1 - l - cap_h) * 4;
    const uint32_t leaf_lens[4] = {cs->leaf_len, nw + (zk ? GL355_SALT_SIZE : 0u), nch * (1 + npp) + (zk ? GL355_SALT_SIZE : 0u), nch * qdf + (zk ? GL355_SALT_SIZE : 0u)};
    uint64_t open_words = 0;
    for (int o = 0; o < 4; o++) open_words += (uint64_t)nq_idx * (leaf_lens[o] + (uint64_t)depth0 * 4);
    open_words += (uint64_t)nq_idx * (L * 4 + sib_total);
    const uint64_t stage_words = (uint64_t)B * std::max<uint64_t>({n_cap * 4, 2 * (n_open_all + nch), 2 * (n >> L) + 8, open_words, 16});
    uint64_t* stage = nullptr;
    GL355_TRY(ctx->pinned(stage_words * 8, reinterpret_cast<void**>(&stage)));

    auto observe_caps = [&](const uint64_t* d_caps) -> int32_t {        // d_caps: [B][n_cap][4]
        GL355_HIP(ctx, ctx->d2h(stage, d_caps, (uint64_t)B * n_cap * 32));
        GL355_HIP(ctx, ctx->wait());
        for (uint32_t u = 0; u < B; u++) {
            memcpy(out[u], stage + (uint64_t)u * n_cap * 4, n_cap * 32);
            gl355_challenger_observe(&ch[u], out[u], n_cap * 4);
            out[u] += n_cap * 4;
        }
        return GL355_OK;
    };

    // ---- wires ------------------------------------------------------------------------------------------------------------------
    BOracle o_w, o_z, o_q;
    GL355_TRY(o_w.alloc(ctx, B, c.degree_bits, c.rate_bits, nw, zk, cap_h));
    Scratch wires_keep(ctx);                    // the wire VALUES are needed again for the permutation argument
    GL355_TRY(wires_keep.get((uint64_t)B * nw * n * 8));
    uint64_t* wires = wires_keep.as<uint64_t>();
    if (d_wires_dense) {
        GL355_HIP(ctx, hipMemcpyAsync(wires, d_wires_dense, (uint64_t)B * nw * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        GL355_HIP(ctx, hipMemsetAsync(wires, 0, (uint64_t)B * nw * n * 8, ctx->stream));
        Scratch rbuf(ctx);
        if (n_rows) {
            const uint64_t per = (uint64_t)n_rows * nw;
            const bool rows_on_device = ptr_is_device(rows_host);       // the batch runtime uploads the next batch's rows on its copy stream
            GL355_TRY(rbuf.get((rows_on_device ? 0 : (uint64_t)B * per * 8) + (uint64_t)n_rows * 4 + 16));
            const uint64_t* d_vals = rows_on_device ? rows_host : rbuf.as<uint64_t>();
            uint32_t* d_idx = reinterpret_cast<uint32_t*>(rbuf.as<uint64_t>() + (rows_on_device ? 0 : (uint64_t)B * per));
            if (!rows_on_device) GL355_HIP(ctx, hipMemcpyAsync(rbuf.p, rows_host, (uint64_t)B * per * 8, hipMemcpyHostToDevice, ctx->stream));
            GL355_HIP(ctx, hipMemcpyAsync(d_idx, row_idx, (uint64_t)n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
            ProfScope ps(ctx, "witness_scatter", (uint64_t)B * per * 16);
            hipLaunchKernelGGL(witness_rows_units_kernel, dim3((uint32_t)((per + 255) / 256), B), dim3(256), 0, ctx->stream, wires, n, nw, d_idx, d_vals, n_rows);
            LAUNCH_CHECK(ctx);
        }
        const uint64_t cnt_b = (uint64_t)n_blind * nw + (uint64_t)n_z_pairs * routed;
        if (cnt_b) {
            ProfScope ps(ctx, "witness_blind", (uint64_t)B * cnt_b * 8);
            hipLaunchKernelGGL(witness_blind_units_kernel, dim3((uint32_t)((cnt_b / 4 + 256) / 256), B), dim3(256), 0, ctx->stream, wires, n, nw, routed, blind_start,
                               n_blind, z_start, n_z_pairs, keys);
            LAUNCH_CHECK(ctx);
        }
        GL355_HIP(ctx, ctx->wait());      // the caller's host row buffers may be reused; rbuf is released after its last use
    }
    GL355_HIP(ctx, hipMemcpyAsync(o_w.coeffs, wires, (uint64_t)B * nw * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    GL355_TRY(commit_units(ctx, hasher, o_w, false, false, &keys, GL355_BLIND_STREAM_WIRES_SALT));
    GL355_TRY(observe_caps(o_w.cap));
    UnitVals betas, gammas, alphas;
    memset(&betas, 0, sizeof betas); memset(&gammas, 0, sizeof gammas); memset(&alphas, 0, sizeof alphas);
    for (uint32_t u = 0; u < B; u++) {
        gl355_challenger_squeeze(&ch[u], &betas.v[u * 4], nch);
        gl355_challenger_squeeze(&ch[u], &gammas.v[u * 4], nch);
    }
    // ---- Z / partial products --------------------------------------------------------------------------------------------------
    const uint32_t z_width = nch * (1 + npp);
    GL355_TRY(o_z.alloc(ctx, B, c.degree_bits, c.rate_bits, z_width, zk, cap_h));
    Staged s_sig(ctx), s_k(ctx);
    GL355_TRY(s_sig.open(pd->sigmas, (uint64_t)routed * n * 8, 1));
    GL355_TRY(s_k.open(pd->k_is, (uint64_t)routed * 8, 1));
    {
        Scratch zs_tmp(ctx);
        const uint32_t n_inst = B * nch;
        GL355_TRY(zs_tmp.get((uint64_t)n_inst * (n_chunks + 1) * n * 8));
        ZsArgs za;
        memset(&za, 0, sizeof za);
        za.wires = wires; za.wires_us = (uint64_t)nw * n; za.sigmas = s_sig.as<uint64_t>(); za.k_is = s_k.as<uint64_t>();
        za.log_n = c.degree_bits; za.n_routed = routed; za.max_degree = qdf; za.nch = nch; za.n_chunks = n_chunks; za.npp = npp;
        za.g = gl_root_of_unity(c.degree_bits);
        za.chunk_q = zs_tmp.as<uint64_t>(); za.row_prod = za.chunk_q + (uint64_t)n_inst * n_chunks * n;
        za.zbuf = o_z.coeffs; za.z_us = (uint64_t)z_width * n;
        for (uint32_t i = 0; i < MAXB * 4; i++) { za.betas.v[i] = gl_canon(betas.v[i]); za.gammas.v[i] = gl_canon(gammas.v[i]); }
        ProfScope ps(ctx, "zs_partial_products", (uint64_t)n_inst * ((uint64_t)routed * n * 16 + (uint64_t)n_chunks * n * 8));
        const uint32_t blocks = (uint32_t)((n + 255) / 256);
        hipLaunchKernelGGL(zs_rows_units_kernel, dim3(blocks, n_inst), dim3(256), 0, ctx->stream, za);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(zs_scan_units_kernel, dim3(n_inst), dim3(1024), 0, ctx->stream, za);
        LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(zs_partials_units_kernel, dim3(blocks, n_inst), dim3(256), 0, ctx->stream, za);
        LAUNCH_CHECK(ctx);
    }
    GL355_TRY(commit_units(ctx, hasher, o_z, false, false, &keys, GL355_BLIND_STREAM_ZS_SALT));
    wires_keep.reset();      // the wire values are not needed any more
    GL355_TRY(observe_caps(o_z.cap));
    for (uint32_t u = 0; u < B; u++) gl355_challenger_squeeze(&ch[u], &alphas.v[u * 4], nch);
    // ---- quotient --------------------------------------------------------------------------------------------------------------------
    GL355_TRY(o_q.alloc(ctx, B, c.degree_bits, c.rate_bits, nch * qdf, zk, cap_h));
    {
        Scratch qv(ctx);
        GL355_TRY(qv.get((uint64_t)B * nch * nq * 8));
        GL355_TRY(quotient_units_dev(ctx, &c, B, cs->lde, o_w.lde, (uint64_t)nw * N, o_z.lde, (uint64_t)z_width * N, N, s_k.as<uint64_t>(), betas.v, gammas.v,
                                     alphas.v, pi_hashes.v, qv.as<uint64_t>()));
        GL355_TRY(intt_from_bitrev_dev(ctx, qv.as<uint64_t>(), nq, o_q.coeffs, nq, c.degree_bits + qdb, B * nch, GL355_COSET_SHIFT));
    }
    GL355_TRY(commit_units(ctx, hasher, o_q, true, true, &keys, GL355_BLIND_STREAM_QUOTIENT_SALT));
    GL355_TRY(observe_caps(o_q.cap));
    UnitVals zetas, fri_alpha;                  // zetas: [u*4+0..1] = zeta, [u*4+2..3] = g * zeta
    memset(&zetas, 0, sizeof zetas); memset(&fri_alpha, 0, sizeof fri_alpha);
    const uint64_t g = gl_root_of_unity(c.degree_bits);
    for (uint32_t u = 0; u < B; u++) {
        gl355_challenger_squeeze(&ch[u], &zetas.v[u * 4], 2);
        zetas.v[u * 4] = gl_canon(zetas.v[u * 4]); zetas.v[u * 4 + 1] = gl_canon(zetas.v[u * 4 + 1]);
        zetas.v[u * 4 + 2] = gl_canon(gl_mul(zetas.v[u * 4], g)); zetas.v[u * 4 + 3] = gl_canon(gl_mul(zetas.v[u * 4 + 1], g));
    }
    // ---- openings (OpeningSet::new): every polynomial at zeta, the Z polynomials at g * zeta ------------------------------------------
    const OView v_cs = view_of(cs), v_w = view_of(o_w), v_z = view_of(o_z), v_q = view_of(o_q);
    PolySet all, zsset;
    memset(&all, 0, sizeof all); memset(&zsset, 0, sizeof zsset);
    const OView* vs[4] = {&v_cs, &v_w, &v_z, &v_q};
    for (int o = 0; o < 4; o++) { all.base[o] = vs[o]->coeffs; all.us[o] = vs[o]->coeffs_us; all.count[o] = vs[o]->batch; }
    all.n_sets = 4; all.log_n = c.degree_bits;
    zsset.base[0] = v_z.coeffs; zsset.us[0] = v_z.coeffs_us; zsset.count[0] = nch; zsset.n_sets = 1; zsset.log_n = c.degree_bits;
    const uint32_t n_open = (uint32_t)n_open_all;
    Scratch evb(ctx);
    GL355_TRY(evb.get((uint64_t)B * (n_open + nch) * 16));
    {
        EvalArgs ea;
        ea.all = all; ea.zs = zsset; ea.n_all = n_open; ea.zeta = zetas; ea.out = evb.as<uint64_t>(); ea.out_us = 2 * (n_open + nch);
        ProfScope ps(ctx, "eval_polys", (uint64_t)B * ((uint64_t)(n_open + nch) << c.degree_bits) * 8);
        hipLaunchKernelGGL(eval_polys_units_kernel, dim3(n_open + nch, B), dim3(256), 0, ctx->stream, ea);
        LAUNCH_CHECK(ctx);
    }
    GL355_HIP(ctx, ctx->d2h(stage, evb.as<uint64_t>(), (uint64_t)B * (n_open + nch) * 16));
    GL355_HIP(ctx, ctx->wait());
    for (uint32_t u = 0; u < B; u++) {
        const uint64_t w = 2ull * (n_open + nch);
        memcpy(out[u], stage + (uint64_t)u * w, w * 8);
        gl355_challenger_observe(&ch[u], out[u], w);
        out[u] += w;
        gl355_challenger_squeeze(&ch[u], &fri_alpha.v[u * 4], 2);
        fri_alpha.v[u * 4] = gl_canon(fri_alpha.v[u * 4]); fri_alpha.v[u * 4 + 1] = gl_canon(fri_alpha.v[u * 4 + 1]);
    }
    // ---- DEEP quotient (prove_openings): acc = Q_zeta * alpha^nch + Q_{g zeta}, two base columns per unit ----------------------------------
    Scratch colsA(ctx), colsB(ctx), fri_vals(ctx);
    GL355_TRY(colsA.get((uint64_t)B * 2 * n * 8));
    GL355_TRY(colsB.get((uint64_t)B * n * 8 + 64));
    GL355_TRY(fri_vals.get((uint64_t)B * 2 * N * 8));
    {
        const uint64_t n_blocks = (n + DEEP_BLK - 1) / DEEP_BLK;
        Scratch tab(ctx), vbuf(ctx);
        GL355_TRY(tab.get((uint64_t)B * deep_tab_len(n_open) * 16));
        GL355_TRY(vbuf.get((uint64_t)B * (2 * n + 4 * n_blocks + 8) * 8));
        DeepTabArgs ta;
        ta.tab = tab.as<uint64_t>(); ta.n_alpha = n_open; ta.alpha = fri_alpha; ta.zeta = zetas;
        hipLaunchKernelGGL(deep_tables_kernel, dim3(B), dim3(256), 0, ctx->stream, ta);
        LAUNCH_CHECK(ctx);
        GL355_HIP(ctx, hipMemsetAsync(colsA.p, 0, (uint64_t)B * 2 * n * 8, ctx->stream));
        DeepArgs da;
        memset(&da, 0, sizeof da);
        da.tab = tab.as<uint64_t>(); da.n_alpha = n_open; da.n = n; da.n_blocks = n_blocks;
        da.v = vbuf.as<uint64_t>(); da.totals = da.v + (uint64_t)B * 2 * n; da.carry = da.totals + (uint64_t)B * 2 * n_blocks;
        da.acc = colsA.as<uint64_t>();
        for (int pass = 0; pass < 2; pass++) {
            da.polys = pass ? zsset : all; da.n_polys = pass ? nch : n_open; da.point = pass;
            ProfScope ps(ctx, "deep_batch", (uint64_t)B * ((uint64_t)da.n_polys * n * 8 + n * 32));
            hipLaunchKernelGGL(deep_reduce_scan_units_kernel, dim3((uint32_t)n_blocks, B), dim3(DEEP_BLK), 0, ctx->stream, da);
            LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(deep_carry_units_kernel, dim3(B), dim3(64), 0, ctx->stream, da);
            LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(deep_finish_units_kernel, dim3((uint32_t)n_blocks, B), dim3(DEEP_BLK), 0, ctx->stream, da);
            LAUNCH_CHECK(ctx);
        }
    }
    // ---- FRI commit phase (fri_committed_trees) ----------------------------------------------------------------------------------------
    std::vector<uint64_t> leaf_off(L), dig_off(L);
    uint64_t tree_words = 0;
    for (uint32_t l = 0; l < L; l++) {
        const uint64_t nl = N >> (l + 1);
        if (nl < n_cap) return ctx->fail(GL355_E_INVALID_ARG, "prove: FRI layer smaller than the cap");
        leaf_off[l] = tree_words; tree_words += (uint64_t)B * nl * 4;
        dig_off[l] = tree_words; tree_words += (uint64_t)B * 2 * (nl - n_cap) * 4;
    }
    Scratch trees(ctx);
    GL355_TRY(trees.get((tree_words + (uint64_t)B * n_cap * 4 + 16) * 8));
    uint64_t* tree_buf = trees.as<uint64_t>();
    uint64_t* d_cap = tree_buf + tree_words;
    uint64_t* cols = colsA.as<uint64_t>();
    uint64_t* cols2 = colsB.as<uint64_t>();
    uint64_t shift = GL355_COSET_SHIFT, len_c = n;
    std::vector<uint64_t*> p_fri_caps(B);
    for (uint32_t u = 0; u < B; u++) { p_fri_caps[u] = out[u]; out[u] += (uint64_t)L * n_cap * 4; }
    for (uint32_t l = 0; l < L; l++) {
        const uint64_t len_v = len_c << c.rate_bits, nl = len_v / 2;
        GL355_TRY(lde_dev(ctx, cols, len_c, log2_u64(len_c), c.rate_bits, gl_canon(shift), 2 * B, fri_vals.as<uint64_t>(), len_v, true));
        uint64_t* lv = tree_buf + leaf_off[l];
        {
            ProfScope ps(ctx, "fri_layer_leaves", (uint64_t)B * len_v * 32);
            hipLaunchKernelGGL(fri_leaves_units_kernel, dim3((uint32_t)((nl + 255) / 256), B), dim3(256), 0, ctx->stream, fri_vals.as<uint64_t>(), len_v, lv);
            LAUNCH_CHECK(ctx);
        }
        LeafArgs la;
        memset(&la, 0, sizeof la);
        la.leaves = lv; la.n_leaves = (uint64_t)B * nl; la.leaf_len = 4; la.col_major = 0; la.stride = 4;
        GL355_TRY(merkle_build_args_any(ctx, hasher, la, log2_u64(nl) - cap_h, tree_buf + dig_off[l], d_cap));
        GL355_HIP(ctx, ctx->d2h(stage, d_cap, (uint64_t)B * n_cap * 32));
        GL355_HIP(ctx, ctx->wait());
        UnitVals beta;
        memset(&beta, 0, sizeof beta);
        for (uint32_t u = 0; u < B; u++) {
            uint64_t* dst = p_fri_caps[u] + (uint64_t)l * n_cap * 4;
            memcpy(dst, stage + (uint64_t)u * n_cap * 4, n_cap * 32);
            gl355_challenger_observe(&ch[u], dst, n_cap * 4);
            gl355_challenger_squeeze(&ch[u], &beta.v[u * 4], 2);
            beta.v[u * 4] = gl_canon(beta.v[u * 4]); beta.v[u * 4 + 1] = gl_canon(beta.v[u * 4 + 1]);
        }
        {
            ProfScope ps(ctx, "fri_fold", (uint64_t)B * (len_c * 16 + len_c * 8));
            hipLaunchKernelGGL(fri_fold_units_kernel, dim3((uint32_t)((len_c / 2 + 255) / 256), B), dim3(256), 0, ctx->stream, cols, len_c, beta, cols2);
            LAUNCH_CHECK(ctx);
        }
        std::swap(cols, cols2);
        len_c >>= 1;
        shift = gl_mul(shift, shift);
    }
    // final polynomial (len_c = n >> L extension coefficients per unit, two columns each)
    GL355_HIP(ctx, ctx->d2h(stage, cols, (uint64_t)B * 2 * len_c * 8));
    GL355_HIP(ctx, ctx->wait());
    PowArgs pa;
    memset(&pa, 0, sizeof pa);
    std::vector<uint64_t*> p_pow(B);
    for (uint32_t u = 0; u < B; u++) {
        const uint64_t* c0 = stage + (uint64_t)u * 2 * len_c;
        for (uint64_t k = 0; k < len_c; k++) { out[u][2 * k] = c0[k]; out[u][2 * k + 1] = c0[len_c + k]; }
        gl355_challenger_observe(&ch[u], out[u], 2 * len_c);
        out[u] += 2 * len_c;
        p_pow[u] = out[u]; out[u] += 1;
        uint64_t st[12];
        uint32_t pos;
        if (gl355_challenger_pow_state(&ch[u], st, &pos) != GL355_OK) return ctx->fail(GL355_E_INVALID_ARG, "prove: challenger state");
        for (int k = 0; k < 12; k++) pa.state[u * 12 + k] = gl_canon(st[k]);
        pa.pos[u] = pos; pa.todo[u] = 1;
    }
    // ---- proof of work (fri_proof_of_work): smallest witness per unit ---------------------------------------------------------------------------
    if (hasher == GL355_HASH_POSEIDON) {
        if (pd->pow_bits > 40) return ctx->fail(GL355_E_UNSUPPORTED, "pow: more than 40 bits of grinding refused");
        Scratch pb(ctx);
        GL355_TRY(pb.get(MAXB * 8));
        pa.best = reinterpret_cast<unsigned long long*>(pb.p);
        pa.bits = pd->pow_bits;
        GL355_HIP(ctx, hipMemsetAsync(pb.p, 0xFF, MAXB * 8, ctx->stream));
        // a launch of 2^(bits+1) candidates holds a solution with probability 1 - e^-2; units without one go again
        uint64_t per_launch = 1ull << std::min<uint32_t>(std::max<uint32_t>(pd->pow_bits + 1, 12), 22);
        uint64_t base = 0;
        for (uint32_t left = B; left;) {
            ProfScope ps(ctx, "pow_grind", 0);
            pa.start = base;
            hipLaunchKernelGGL(pow_grind_units_kernel, dim3((uint32_t)(per_launch / 256), B), dim3(256), 0, ctx->stream, pa);
            LAUNCH_CHECK(ctx);
            GL355_HIP(ctx, ctx->d2h(stage, pb.p, (uint64_t)B * 8));
            GL355_HIP(ctx, ctx->wait());
            for (uint32_t u = 0; u < B; u++)
                if (pa.todo[u] && stage[u] != ~0ull) { *p_pow[u] = stage[u]; pa.todo[u] = 0; left--; }
            base += per_launch;
            if (per_launch < (1ull << 22)) per_launch <<= 1;
            if (base > (1ull << 44)) return ctx->fail(GL355_E_UNSUPPORTED, "pow: no witness found in 2^44 candidates");
        }
    } else {
        for (uint32_t u = 0; u < B; u++) GL355_TRY(pow_grind_any(ctx, hasher, &pa.state[u * 12], pa.pos[u], pd->pow_bits, 0, p_pow[u]));
    }
    // ---- query indices, then every opening of every query in five launches -----------------------------------------------------------------------
    std::vector<uint64_t> q_idx((uint64_t)B * nq_idx);
    for (uint32_t u = 0; u < B; u++) {
        gl355_challenger_observe(&ch[u], p_pow[u], 1);
        uint64_t resp;
        gl355_challenger_squeeze(&ch[u], &resp, 1);
        if (pd->pow_bits && (resp >> (64 - pd->pow_bits)) != 0) return ctx->fail(GL355_E_HIP, "prove: proof-of-work response check failed");
        gl355_challenger_squeeze(&ch[u], &q_idx[(uint64_t)u * nq_idx], nq_idx);
        for (uint32_t q = 0; q < nq_idx; q++) q_idx[(uint64_t)u * nq_idx + q] &= (N - 1);
    }
    Scratch ob(ctx);
    GL355_TRY(ob.get(((uint64_t)B * nq_idx + (uint64_t)B * open_words + 16) * 8));
    uint64_t* d_idx = ob.as<uint64_t>();
    uint64_t* d_open = d_idx + (uint64_t)B * nq_idx;
    GL355_HIP(ctx, hipMemcpyAsync(d_idx, q_idx.data(), (uint64_t)B * nq_idx * 8, hipMemcpyHostToDevice, ctx->stream));
    uint64_t off_leaf[4], off_sib[4], woff = 0;
    for (int o = 0; o < 4; o++) {
        off_leaf[o] = woff; woff += (uint64_t)B * nq_idx * leaf_lens[o];
        off_sib[o] = woff; woff += (uint64_t)B * nq_idx * depth0 * 4;
        OpenArgs oa;
        oa.o = *vs[o]; oa.N = N; oa.lde_bits = lde_bits; oa.cap_height = cap_h; oa.n_idx = nq_idx; oa.idx = d_idx;
        oa.leaves = d_open + off_leaf[o]; oa.sibs = d_open + off_sib[o];
        if (oa.o.leaf_len != leaf_lens[o]) return ctx->fail(GL355_E_HIP, "prove: internal leaf-length mismatch");
        ProfScope ps(ctx, "open_batch", (uint64_t)B * nq_idx * (leaf_lens[o] * 16 + depth0 * 64));
        hipLaunchKernelGGL(open_units_kernel, dim3(nq_idx, B), dim3(64), 0, ctx->stream, oa);
        LAUNCH_CHECK(ctx);
    }
    const uint64_t off_ev = woff; woff += (uint64_t)B * nq_idx * L * 4;
    const uint64_t off_fsib = woff; woff += (uint64_t)B * nq_idx * sib_total;
    if (L) {
        OpenFriArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.trees = tree_buf;
        uint64_t so = 0;
        for (uint32_t l = 0; l < L; l++) { fa.leaf_off[l] = leaf_off[l]; fa.dig_off[l] = dig_off[l]; fa.sib_off[l] = so; so += (uint64_t)(lde_bits - 1 - l - cap_h) * 4; }
        fa.sib_total = sib_total; fa.lde_bits = lde_bits; fa.cap_height = cap_h; fa.n_idx = nq_idx; fa.n_layers = L; fa.idx = d_idx;
        fa.evals = d_open + off_ev; fa.sibs = d_open + off_fsib;
        ProfScope ps(ctx, "open_batch", (uint64_t)B * nq_idx * (L * 64 + sib_total * 16));
        hipLaunchKernelGGL(open_fri_units_kernel, dim3(nq_idx, B, L), dim3(64), 0, ctx->stream, fa);
        LAUNCH_CHECK(ctx);
    }
    GL355_HIP(ctx, ctx->d2h(stage, d_open, woff * 8));
    GL355_HIP(ctx, ctx->wait());
    for (uint32_t u = 0; u < B; u++) {
        uint64_t* o_ = out[u];
        for (uint32_t q = 0; q < nq_idx; q++) {
            *o_++ = q_idx[(uint64_t)u * nq_idx + q];
            for (int o = 0; o < 4; o++) {
                const uint32_t ll = leaf_lens[o];
                memcpy(o_, stage + off_leaf[o] + ((uint64_t)u * nq_idx + q) * ll, (uint64_t)ll * 8); o_ += ll;
                memcpy(o_, stage + off_sib[o] + ((uint64_t)u * nq_idx + q) * depth0 * 4, (uint64_t)depth0 * 32); o_ += (uint64_t)depth0 * 4;
            }
            uint64_t so = 0;
            for (uint32_t l = 0; l < L; l++) {
                memcpy(o_, stage + off_ev + (((uint64_t)u * nq_idx + q) * L + l) * 4, 32); o_ += 4;
                const uint64_t d = (uint64_t)(lde_bits - 1 - l - cap_h) * 4;
                memcpy(o_, stage + off_fsib + ((uint64_t)u * nq_idx + q) * sib_total + so, d * 8); o_ += d;
                so += d;
            }
        }
        if ((uint64_t)(o_ - io[u].proof) != need) return ctx->fail(GL355_E_HIP, "prove: internal proof-size mismatch");
    }
    return GL355_OK;
}

}  // namespace gl355
