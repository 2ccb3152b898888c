// DEEP quotient, openings, FRI fold / layer commit, permutation-argument Z and the element-wise field
// entry points for gfx950 (a1, a9, a11, a12 of SURVEY.md 8).
//
// Replaces PolynomialBatch::prove_openings (ReducingFactor::reduce_polys_base,
// PolynomialCoeffs::divide_by_linear, shift_poly), OpeningSet::new's polynomial evaluations,
// fri_committed_trees' reduce_with_powers fold and wires_permutation_partial_products_and_zs,
// all reached from the reference through CircuitData::prove (src/plonky2_semaphore/access_set.rs:94,
// recursion.rs:168, wrapper.rs:55).  Formulas pinned by the reference's verifier:
// chip/fri_chip.rs:112-149 (batch combine), :168-226 (arity-2 fold), types/fri.rs:50-73 (batches),
// chip/plonk/vanishing_poly.rs:54-108,183-218 (Z / partial products).
//
// These are streaming kernels: every polynomial coefficient is read exactly once, lane k handles
// coefficient k so each wave reads 512 contiguous bytes per column.
#include "gl355_internal.h"

namespace gl355 {

// ---- a1: element-wise field ops (test / utility surface) -------------------------------------
__global__ void field_batch_kernel(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
        case GL355_OP_ADD: out[i] = gl_canon(gl_add(a[i], b[i])); break;
        case GL355_OP_SUB: out[i] = gl_canon(gl_sub(a[i], b[i])); break;
        case GL355_OP_MUL: out[i] = gl_canon(gl_mul(a[i], b[i])); break;
        case GL355_OP_INV: out[i] = gl_canon(gl_inv(a[i])); break;
        case GL355_OP_EXT_MUL: {
            gl2 r = gl2_canon(gl2_mul(gl2_make(a[2 * i], a[2 * i + 1]), gl2_make(b[2 * i], b[2 * i + 1])));
            out[2 * i] = r.c0; out[2 * i + 1] = r.c1;
        } break;
        case GL355_OP_EXT_INV: {
            gl2 r = gl2_canon(gl2_inv(gl2_make(a[2 * i], a[2 * i + 1])));
            out[2 * i] = r.c0; out[2 * i + 1] = r.c1;
        } break;
    }
}
int32_t field_batch_dev(Ctx* ctx, int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n) {
    if (n == 0) return GL355_OK;
    hipLaunchKernelGGL(field_batch_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, a, b, out, n);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

__global__ void canon_kernel(uint64_t* a, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) a[i] = gl_canon(a[i]);
}
int32_t canon_dev(Ctx* ctx, uint64_t* a, uint64_t n) {
    if (n == 0) return GL355_OK;
    hipLaunchKernelGGL(canon_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, n);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ---- a11: DEEP quotient -----------------------------------------------------------------------
constexpr int DEEP_B = 256;  // coefficients per workgroup in the division scan

// phase 1: comp[k] = sum_i alpha^i p_i[k]; then in-block suffix Horner sums v[k] = sum_{j>=k, j in
// block} comp[j] z^(j-k) by a log-step scan with multipliers z^(2^s); block totals to `totals`.
__global__ void __launch_bounds__(DEEP_B) deep_reduce_scan_kernel(const uint64_t* const* polys, uint32_t n_polys,
                                                                  const uint64_t* alpha_pows /* n_polys ext */,
                                                                  const uint64_t* z_pow2 /* z^(2^s), s < 8, ext */,
                                                                  uint64_t n, uint64_t* v_out, uint64_t* totals) {
    __shared__ uint64_t sh[2 * DEEP_B];
    const int tid = threadIdx.x;
    const uint64_t k = blockIdx.x * (uint64_t)DEEP_B + tid;
    gl2 c = gl2_make(0, 0);
    if (k < n) {
        for (uint32_t i = 0; i < n_polys; i++) {
            const uint64_t coef = polys[i][k];
            c.c0 = gl_add(c.c0, gl_mul(alpha_pows[2 * i], coef));
            c.c1 = gl_add(c.c1, gl_mul(alpha_pows[2 * i + 1], coef));
        }
    }
    sh[2 * tid] = c.c0; sh[2 * tid + 1] = c.c1;
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 8; s++) {
        const int other = tid + (1 << s);
        gl2 add = gl2_make(0, 0);
        if (other < DEEP_B) add = gl2_mul(gl2_make(sh[2 * other], sh[2 * other + 1]), gl2_make(z_pow2[2 * s], z_pow2[2 * s + 1]));
        __syncthreads();
        c = gl2_add(c, add);
        sh[2 * tid] = c.c0; sh[2 * tid + 1] = c.c1;
        __syncthreads();
    }
    if (k < n) { v_out[2 * k] = c.c0; v_out[2 * k + 1] = c.c1; }
    if (tid == 0) { totals[2 * blockIdx.x] = c.c0; totals[2 * blockIdx.x + 1] = c.c1; }
}

// phase 2 (one lane): carry[blk] = b_{(blk+1) B} = T_{blk+1} + z^B carry[blk+1]
__global__ void deep_carry_kernel(const uint64_t* totals, uint64_t n_blocks, const uint64_t* z_pow_b, uint64_t* carry) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const gl2 zb = gl2_make(z_pow_b[0], z_pow_b[1]);
    gl2 c = gl2_make(0, 0);
    for (uint64_t blk = n_blocks; blk-- > 0;) {
        carry[2 * blk] = c.c0; carry[2 * blk + 1] = c.c1;
        c = gl2_add(gl2_make(totals[2 * blk], totals[2 * blk + 1]), gl2_mul(zb, c));
    }
}

// phase 3: quotient q_k = b_{k+1}, b_j = v[j] + z^(B - j%B) carry[blk(j)]; acc = acc*shift + q
__global__ void __launch_bounds__(DEEP_B) deep_finish_kernel(const uint64_t* v, const uint64_t* carry,
                                                             const uint64_t* z_pows /* z^0..z^B ext */, uint64_t n,
                                                             const uint64_t* shift /* alpha^n_polys */, uint64_t* acc) {
    const uint64_t k = blockIdx.x * (uint64_t)DEEP_B + threadIdx.x;
    if (k >= n) return;
    gl2 q = gl2_make(0, 0);
    const uint64_t j = k + 1;
    if (j < n) {
        const uint64_t blk = j / DEEP_B, off = j % DEEP_B;
        const gl2 cr = gl2_make(carry[2 * blk], carry[2 * blk + 1]);
        const uint64_t e = DEEP_B - off;
        q = gl2_add(gl2_make(v[2 * j], v[2 * j + 1]), gl2_mul(gl2_make(z_pows[2 * e], z_pows[2 * e + 1]), cr));
    }
    const gl2 a = gl2_mul(gl2_make(acc[2 * k], acc[2 * k + 1]), gl2_make(shift[0], shift[1]));
    const gl2 r = gl2_canon(gl2_add(a, q));
    acc[2 * k] = r.c0; acc[2 * k + 1] = r.c1;
}

int32_t deep_batch_dev(Ctx* ctx, const uint64_t* const* poly_ptrs_host, uint32_t n_polys, uint32_t log_n,
                       const uint64_t alpha[2], const uint64_t z[2], uint64_t* acc) {
    const uint64_t n = 1ull << log_n;
    const uint64_t n_blocks = (n + DEEP_B - 1) / DEEP_B;
    // small host-computed tables: alpha^i (i < n_polys), alpha^n_polys, z^(2^s), z^0..z^B
    std::vector<uint64_t> host;
    host.reserve(2 * (n_polys + 1 + 8 + DEEP_B + 1));
    gl2 al = gl2_canon(gl2_make(alpha[0], alpha[1])), zz = gl2_canon(gl2_make(z[0], z[1]));
    gl2 ap = gl2_make(1, 0);
    for (uint32_t i = 0; i <= n_polys; i++) { gl2 c = gl2_canon(ap); host.push_back(c.c0); host.push_back(c.c1); ap = gl2_mul(ap, al); }
    gl2 zp = zz;
    for (int s = 0; s < 8; s++) { gl2 c = gl2_canon(zp); host.push_back(c.c0); host.push_back(c.c1); zp = gl2_mul(zp, zp); }
    gl2 zk = gl2_make(1, 0);
    for (int e = 0; e <= DEEP_B; e++) { gl2 c = gl2_canon(zk); host.push_back(c.c0); host.push_back(c.c1); zk = gl2_mul(zk, zz); }
    const size_t tab_u64 = host.size();
    const size_t ptr_bytes = sizeof(uint64_t*) * n_polys;
    Scratch tabs(ctx), vbuf(ctx);
    GL355_TRY(tabs.get(tab_u64 * 8 + ptr_bytes + 16));
    GL355_TRY(vbuf.get((2 * n + 4 * n_blocks + 8) * 8));
    uint64_t* d_tab = tabs.as<uint64_t>();
    const uint64_t** d_ptrs = reinterpret_cast<const uint64_t**>(d_tab + tab_u64);
    GL355_HIP(ctx, hipMemcpyAsync(d_tab, host.data(), tab_u64 * 8, hipMemcpyHostToDevice, ctx->stream));
    GL355_HIP(ctx, hipMemcpyAsync(d_ptrs, poly_ptrs_host, ptr_bytes, hipMemcpyHostToDevice, ctx->stream));
    // the host vectors must outlive the async copies
    GL355_HIP(ctx, ctx->wait());
    const uint64_t* d_alpha = d_tab;
    const uint64_t* d_shift = d_tab + 2 * n_polys;
    const uint64_t* d_zpow2 = d_tab + 2 * (n_polys + 1);
    const uint64_t* d_zpows = d_zpow2 + 16;
    uint64_t* d_v = vbuf.as<uint64_t>();
    uint64_t* d_tot = d_v + 2 * n;
    uint64_t* d_carry = d_tot + 2 * n_blocks;
    ProfScope ps(ctx, "deep_batch", (uint64_t)n_polys * n * 8 + n * 32);   // every coefficient once + acc in/out
    hipLaunchKernelGGL(deep_reduce_scan_kernel, dim3((uint32_t)n_blocks), dim3(DEEP_B), 0, ctx->stream, d_ptrs, n_polys,
                       d_alpha, d_zpow2, n, d_v, d_tot);
    GL355_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(deep_carry_kernel, dim3(1), dim3(64), 0, ctx->stream, d_tot, n_blocks, d_zpows + 2 * DEEP_B, d_carry);
    GL355_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(deep_finish_kernel, dim3((uint32_t)n_blocks), dim3(DEEP_B), 0, ctx->stream, d_v, d_carry, d_zpows, n,
                       d_shift, acc);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ---- openings: p_i(z) for base-field coefficient columns, z in F_p^2 ---------------------------
// one workgroup per polynomial; lane t Horner-evaluates the coefficients k = t (mod 256) in z^256 (coalesced loads) and scales
// by z^t; the partials are summed in LDS.
__global__ void __launch_bounds__(256) eval_polys_kernel(const uint64_t* const* polys, uint64_t n, const uint64_t* z,
                                                        uint64_t* out) {
    __shared__ uint64_t sh[512];
    const int tid = threadIdx.x;
    const uint64_t* p = polys[blockIdx.x];
    const gl2 zz = gl2_make(z[0], z[1]);
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
    if (tid == 0) { out[2 * blockIdx.x] = gl_canon(sh[0]); out[2 * blockIdx.x + 1] = gl_canon(sh[1]); }
}

int32_t eval_polys_ext_dev(Ctx* ctx, const uint64_t* const* poly_ptrs_host, uint32_t n_polys, uint32_t log_n,
                           const uint64_t z[2], uint64_t* out_dev) {
    if (n_polys == 0) return GL355_OK;
    Scratch sc(ctx);
    GL355_TRY(sc.get(sizeof(uint64_t*) * n_polys + 16));
    uint64_t* d_z = sc.as<uint64_t>();
    const uint64_t** d_ptrs = reinterpret_cast<const uint64_t**>(d_z + 2);
    uint64_t zc[2] = {gl_canon(z[0]), gl_canon(z[1])};
    GL355_HIP(ctx, hipMemcpyAsync(d_z, zc, 16, hipMemcpyHostToDevice, ctx->stream));
    GL355_HIP(ctx, hipMemcpyAsync(d_ptrs, poly_ptrs_host, sizeof(uint64_t*) * n_polys, hipMemcpyHostToDevice, ctx->stream));
    GL355_HIP(ctx, ctx->wait());
    ProfScope ps(ctx, "eval_polys", ((uint64_t)n_polys << log_n) * 8);
    hipLaunchKernelGGL(eval_polys_kernel, dim3(n_polys), dim3(256), 0, ctx->stream, d_ptrs, 1ull << log_n, d_z, out_dev);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ---- a12: FRI ---------------------------------------------------------------------------------
__global__ void fri_fold_kernel(const uint64_t* c, uint64_t half, uint64_t b0, uint64_t b1, uint64_t* out) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= half) return;
    const ulonglong2 e = *reinterpret_cast<const ulonglong2*>(c + 4 * k);
    const ulonglong2 o = *reinterpret_cast<const ulonglong2*>(c + 4 * k + 2);
    const gl2 r = gl2_canon(gl2_add(gl2_make(e.x, e.y), gl2_mul(gl2_make(o.x, o.y), gl2_make(b0, b1))));
    *reinterpret_cast<ulonglong2*>(out + 2 * k) = make_ulonglong2(r.c0, r.c1);
}
int32_t fri_fold_dev(Ctx* ctx, const uint64_t* coeffs, uint64_t n, const uint64_t beta[2], uint64_t* out) {
    const uint64_t half = n / 2;
    if (half == 0) return GL355_OK;
    ProfScope ps(ctx, "fri_fold", n * 16 + half * 16);
    hipLaunchKernelGGL(fri_fold_kernel, dim3((uint32_t)((half + 255) / 256)), dim3(256), 0, ctx->stream, coeffs, half,
                       gl_canon(beta[0]), gl_canon(beta[1]), out);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// leaves[i] = (v[br(2i)], v[br(2i+1)]) -- i.e. ext element j of the bit-reversed sequence
__global__ void fri_layer_leaves_kernel(const uint64_t* values, uint32_t log_n, uint64_t* leaves) {
    const uint64_t n = 1ull << log_n;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = log_n ? (__brevll(i) >> (64 - log_n)) : 0;
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(values + 2 * j);
    *reinterpret_cast<ulonglong2*>(leaves + 2 * i) = make_ulonglong2(gl_canon(v.x), gl_canon(v.y));
}
int32_t fri_layer_leaves_dev(Ctx* ctx, const uint64_t* values, uint64_t n, uint64_t* leaves) {
    if (n == 0) return GL355_OK;
    ProfScope ps(ctx, "fri_layer_leaves", n * 32);
    hipLaunchKernelGGL(fri_layer_leaves_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, values,
                       log2_u64(n), leaves);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ---- ext <-> two base columns (the F_p^2 LDE is two base-field LDEs) -------------------------------
__global__ void ext_split_kernel(const uint64_t* ext, uint64_t n, uint64_t* c0, uint64_t* c1) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(ext + 2 * k);
    c0[k] = v.x; c1[k] = v.y;
}
__global__ void ext_join_kernel(const uint64_t* c0, const uint64_t* c1, uint64_t n, uint64_t* ext) {
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    *reinterpret_cast<ulonglong2*>(ext + 2 * k) = make_ulonglong2(c0[k], c1[k]);
}
int32_t lde_ext_dev(Ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift, uint64_t* out,
                    bool out_bitrev) {
    const uint64_t n = 1ull << log_n, N = n << rate_bits;
    Scratch sc(ctx);
    GL355_TRY(sc.get((2 * n + 2 * N) * 8));
    uint64_t* cols = sc.as<uint64_t>();
    uint64_t* res = cols + 2 * n;
    { ProfScope ps(ctx, "ext_split_join", n * 32);
    hipLaunchKernelGGL(ext_split_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, coeffs, n, cols, cols + n);
    GL355_HIP(ctx, hipGetLastError()); }
    GL355_TRY(lde_dev(ctx, cols, n, log_n, rate_bits, shift, 2, res, N, out_bitrev));
    ProfScope ps(ctx, "ext_split_join", N * 32);
    hipLaunchKernelGGL(ext_join_kernel, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, ctx->stream, res, res + N, N, out);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ---- a9: permutation argument ---------------------------------------------------------------------
// kernel 1: per row, the n_chunks chunk quotients prod(num)/prod(den) and their product
__global__ void zs_rows_kernel(const uint64_t* wires, const uint64_t* sigmas, const uint64_t* k_is, uint32_t log_n,
                               uint32_t n_routed, uint32_t max_degree, uint64_t beta, uint64_t gamma, uint64_t g,
                               uint64_t* chunk_q /* [n_chunks][n] */, uint64_t* row_prod /* [n] */) {
    const uint64_t n = 1ull << log_n;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = gl_pow(g, i);
    const uint64_t bx = gl_mul(beta, x);
    const uint32_t n_chunks = (n_routed + max_degree - 1) / max_degree;
    uint64_t rp = 1;
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        uint64_t num = 1, den = 1;
        for (uint32_t j = ch * max_degree; j < (ch + 1) * max_degree && j < n_routed; j++) {
            const uint64_t w = wires[(uint64_t)j * n + i];
            num = gl_mul(num, gl_add(gl_add(w, gl_mul(bx, k_is[j])), gamma));
            den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, sigmas[(uint64_t)j * n + i])), gamma));
        }
        const uint64_t q = gl_mul(num, gl_inv(den));
        chunk_q[(uint64_t)ch * n + i] = q;
        rp = gl_mul(rp, q);
    }
    row_prod[i] = rp;
}
// kernel 2 (one workgroup): z[i] = prod_{j<i} row_prod[j], tile-by-tile log-step product scan
__global__ void __launch_bounds__(1024) zs_scan_kernel(const uint64_t* row_prod, uint64_t n, uint64_t* z) {
    __shared__ uint64_t sh[1024];
    const int tid = threadIdx.x;
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
        // v = inclusive product of the tile up to tid; exclusive = product up to tid-1
        const uint64_t excl = tid ? sh[tid - 1] : 1;
        if (i < n) z[i] = gl_canon(gl_mul(running, excl));
        const uint64_t tile_total = sh[1023];
        __syncthreads();
        running = gl_mul(running, tile_total);
    }
}
// kernel 3: partial products: acc = z[i]; acc *= q[ch][i]; pp[ch][i] = acc  (ch < n_chunks-1)
__global__ void zs_partials_kernel(const uint64_t* z, const uint64_t* chunk_q, uint64_t n, uint32_t n_chunks, uint64_t* pp) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t acc = z[i];
    for (uint32_t ch = 0; ch + 1 < n_chunks; ch++) {
        acc = gl_mul(acc, chunk_q[(uint64_t)ch * n + i]);
        pp[(uint64_t)ch * n + i] = gl_canon(acc);
    }
}
int32_t zs_partial_products_dev(Ctx* ctx, const uint64_t* wires, const uint64_t* sigmas, const uint64_t* k_is,
                                uint32_t log_n, uint32_t n_routed, uint32_t max_degree, uint64_t beta, uint64_t gamma,
                                uint64_t* z_out, uint64_t* pp_out) {
    const uint64_t n = 1ull << log_n;
    const uint32_t n_chunks = (n_routed + max_degree - 1) / max_degree;
    Scratch sc(ctx);
    GL355_TRY(sc.get(((uint64_t)n_chunks + 1) * n * 8));
    uint64_t* d_q = sc.as<uint64_t>();
    uint64_t* d_rp = d_q + (uint64_t)n_chunks * n;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    ProfScope ps(ctx, "zs_partial_products", (uint64_t)n_routed * n * 16 + (uint64_t)n_chunks * n * 8);
    hipLaunchKernelGGL(zs_rows_kernel, dim3(blocks), dim3(256), 0, ctx->stream, wires, sigmas, k_is, log_n, n_routed,
                       max_degree, gl_canon(beta), gl_canon(gamma), gl_root_of_unity(log_n), d_q, d_rp);
    GL355_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(zs_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_rp, n, z_out);
    GL355_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(zs_partials_kernel, dim3(blocks), dim3(256), 0, ctx->stream, z_out, d_q, n, n_chunks, pp_out);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

}  // namespace gl355
