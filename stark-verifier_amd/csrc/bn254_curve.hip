// SURVEY 8(f) N4, first slice: the two kernels of the reference's SNARK finalisation on the GPU.
//
// `verify_inside_snark` (src/plonky2_verifier/verifier_api.rs:57-96) runs ParamsKZG::setup (:77), keygen_vk / keygen_pk (:78-79) and
// create_proof (:90) of halo2_proofs at k = 20..23 (chip/native_chip/test_utils.rs:57-95; README: ~505 s, the largest wall-time
// item of the product).  Inside, almost all of the time is two primitives of `halo2_proofs::arithmetic` over halo2curves' bn256:
//   best_fft       radix-2 FFT over the scalar field Fr (2-adicity 28, ROOT_OF_UNITY = 7^((r-1)/2^28))          -> gl355_bn254_fr_ntt
//   best_multiexp  multi-scalar multiplication over G1: y^2 = x^3 + 3 over Fq, sum_i scalars[i] * bases[i]      -> gl355_bn254_g1_msm
// halo2_proofs / halo2curves are un-vendored dependencies; the kernels follow their published definitions and are checked against
// oracle/bn254_curve_oracle.c (itself pinned by halo2curves' ROOT_OF_UNITY and the EIP-196 2*G vector).
//
// Arithmetic: 8 x 32-bit limbs, Montgomery form with R = 2^256, CIOS on v_mad_u64_u32 as in bn254.cuh (4m < R for both primes, so
// products of operands < 2m stay < 2m without a final subtraction; sums / differences take one conditional subtraction of 2m).
// MSM: Pippenger's bucket method, bucket-parallel: per window a counting sort of the point indices by digit (histogram with
// atomics, scan, scatter), one lane per bucket accumulating its points with mixed Jacobian + affine additions, a chunked running-sum
// reduction of the buckets of a window, and a Horner combination of the windows.  This is a correct first slice with the right
// structure for the hardware (integer VALU bound like everything else here), not yet a tuned one: no signed digits, no batched
// affine additions, FFT stages one global pass each.
#include "bn254_field.cuh"
#include "bn254_f29.cuh"
#include <memory>
#include <vector>

// gl355_bn254_g1_msm_prepare's handle: the window multiples of a base set (device), see msm_table_build_kernel
struct gl355_msm_bases {
    gl355::Ctx* ctx;
    uint32_t* tab;              // [wps][n][16]
    uint64_t n;
    uint32_t c, wps;
};
namespace gl355 {

// host_bn254_curve.cpp: sum_w 2^(c w) (S_w + Wt_w) as an affine point (canonical integers; zeros = the identity)
void bn254_g1_horner_host(const uint32_t* s, const uint32_t* wt, uint32_t n_windows, uint32_t c, uint64_t result[8]);

// ================================================================ Fr FFT ===========================================
// values live in Montgomery form between the conversion kernels
__global__ void fr_to_mont_kernel(uint64_t* data, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store256(data + 4 * i, m_from_int<F_R>(load256(data + 4 * i)));
}
// out of Montgomery form, optionally times `scale` (a plain integer, e.g. n^-1: x R * s * R^-1 = x s)
__global__ void fr_from_mont_kernel(uint64_t* data, uint64_t n, u256 scale, int use_scale) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u256 x = load256(data + 4 * i);
    store256(data + 4 * i, use_scale ? m_canon<F_R>(m_mul<F_R>(x, scale)) : m_to_int<F_R>(x));
}
// tw[i] = w^i (Montgomery), i < count, in two steps: lo[j] = w^j (j < 1024) and hi[j] = w^(1024 j) by exponentiation (a few thousand
// entries), then one product per entry.  (One exponentiation per entry -- ~30 products each -- cost more than the transform it served:
// 1.5e7 against 1.0e7 field products at k = 20.)
__global__ void fr_twiddle_seed_kernel(uint64_t* lo, uint64_t* hi, uint64_t n_hi, u256 w_mont) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < 1024) store256(lo + 4 * i, m_pow_u64<F_R>(w_mont, i));
    else if (i - 1024 < n_hi) store256(hi + 4 * (i - 1024), m_pow_u64<F_R>(w_mont, (i - 1024) << 10));
}
__global__ void fr_twiddle_kernel(uint64_t* tw, uint64_t count, const uint64_t* lo, const uint64_t* hi) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    store256(tw + 4 * i, m_mul<F_R>(load256(lo + 4 * (i & 1023)), load256(hi + 4 * (i >> 10))));
}
// the same powers as PLAIN integers times a plain factor f: (lo hi) R * f * R^-1 = lo hi f
__global__ void fr_power_plain_kernel(uint64_t* tab, uint64_t count, const uint64_t* lo, const uint64_t* hi, u256 f) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    store256(tab + 4 * i, m_canon<F_R>(m_mul<F_R>(m_mul<F_R>(load256(lo + 4 * (i & 1023)), load256(hi + 4 * (i >> 10))), f)));
}
// the same powers in Montgomery form times a Montgomery factor
__global__ void fr_power_mont_kernel(uint64_t* tab, uint64_t count, const uint64_t* lo, const uint64_t* hi, u256 f_mont) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    store256(tab + 4 * i, m_mul<F_R>(m_mul<F_R>(load256(lo + 4 * (i & 1023)), load256(hi + 4 * (i >> 10))), f_mont));
}
// out[i] = in[i] (* post[i]) (* scale): the tail of a one-pass in-place resident transform
__global__ void fr_scale_copy_kernel(const uint64_t* in, uint64_t* out, uint64_t n, const uint64_t* post, u256 scale, int use_scale) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u256 x = load256(in + 4 * i);
    if (post) x = m_mul<F_R>(x, load256(post + 4 * i));
    else if (use_scale) x = m_mul<F_R>(x, scale);
    store256(out + 4 * i, x);
}
__global__ void fr_bitrev_kernel(uint64_t* data, uint32_t log_n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= (1ull << log_n)) return;
    const uint64_t j = __brevll(i) >> (64 - log_n);
    if (i < j) {
        const u256 a = load256(data + 4 * i), b = load256(data + 4 * j);
        store256(data + 4 * i, b);
        store256(data + 4 * j, a);
    }
}
// decimation-in-time stage s (span m = 2^s) after the bit reversal: one butterfly per lane
__global__ void __launch_bounds__(256) fr_stage_kernel(uint64_t* data, const uint64_t* tw, uint32_t log_n, uint32_t s) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= (1ull << (log_n - 1))) return;
    const uint64_t half = 1ull << (s - 1);
    const uint64_t j = t & (half - 1), base = (t >> (s - 1)) << s;
    uint64_t* pu = data + 4 * (base + j);
    uint64_t* pv = pu + 4 * half;
    const u256 w = load256(tw + 4 * (j << (log_n - s)));        // w_m^j = w_n^(j n / m)
    const u256 u = load256(pu), v = m_mul<F_R>(load256(pv), w);
    store256(pu, m_add<F_R>(u, v));
    store256(pv, m_sub<F_R>(u, v));
}

// the block constants of the coset form (FrPass::btw): out[2^(t-1) + b] = gpow[t] * w_(2^t)^rev(b), 1 <= t <= log_n, b < 2^(t-1); gpow[t] = g^(n / 2^t)
// (Montgomery, from the host), w_(2^t)^r = tw[r 2^(log_n - t)].  Entry 0 is unused.
__global__ void fr_coset_twiddle_kernel(const uint64_t* tw, const uint64_t* gpow, uint32_t log_n, uint64_t* out) {
    const uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (e == 0 || e >= (1ull << log_n)) return;
    const uint32_t t = 64 - __clzll((unsigned long long)e);                   // 2^(t-1) <= e < 2^t
    const uint64_t b = e - (1ull << (t - 1));
    const uint64_t r = t > 1 ? __brevll(b) >> (64 - (t - 1)) : 0;
    store256(out + 4 * e, m_mul<F_R>(load256(tw + 4 * (r << (log_n - t))), load256(gpow + 4 * t)));
}
// Several decimation-in-time stages per pass over HBM: a workgroup takes a tile of 2^ns "rows" at stride 2^s0 times C adjacent
// columns (1024 elements, 32 KB of LDS as 8 limb planes so that lanes hit consecutive banks) through stages s0+1 .. s0+ns.  The first
// pass (s0 = 0: contiguous 1024-element blocks, ten stages) also does the bit reversal and the conversion to Montgomery form on its
// loads, the last one the conversion back (and the 1/n of the inverse) on its stores: k = 20 is three passes (10 + 5 + 5 stages)
// instead of 23 (conversion, bit reversal, 20 stages, conversion).
struct FrPass {
    const uint64_t* in;         // first pass: the caller's data (natural order, plain integers); later passes: == out
    uint64_t* out;
    const uint64_t* tw;         // w_n^k, k < n / 2, Montgomery form
    uint32_t log_n, s0, ns;
    uint32_t first, last, use_scale;
    u256 scale;
    // coset forms (coeff_to_extended / extended_to_coeff): the first pass reads n_in <= n inputs (zero beyond) times pre[i] (Montgomery),
    // the last pass writes n_out <= n outputs times post[i] (plain integers, the 1/n included) instead of `scale`
    uint64_t n_in, n_out;
    const uint64_t* pre;
    const uint64_t* post;
    // resident form (the PLONK prover, plonk_bn254.hip): inputs already in Montgomery form / outputs left in it (post and scale are then
    // Montgomery values too)
    uint32_t in_mont, out_mont;
    // no_gather: the first pass reads element i where the classic form reads bitrev(i) -- either the input already is in bit-reversed order
    // (decimation in time then gives natural-order output without the scattered 32-byte reads: 1.05 of the 1.85 ms of a 2^23-point transform
    // went into that one pass), or `dif` is set: decimation in frequency, natural-order input, stages from the top down with the butterfly
    // (u + v, (u - v) w), bit-reversed output.  The passes of a dif transform run from the highest s0 down; `first` then marks the pass that
    // reads the input (conversion, pre-multipliers, zero padding), `last` the s0 = 0 pass.
    uint32_t no_gather, dif;
    // COSET form (bn254_fr_ntt_mont_coset_dif): out[bitrev(k)] = sum_i in[i] g^i w^(ik) with NO multiplication by g^i up front and no per-position
    // twiddles.  Write a block of 2^s values as the coset transform of its own shift g': its halves u, v combine as (u + G v, u - G v) with the ONE
    // constant G = g'^(2^(s-1)) per block, and the halves are coset transforms again, with shifts g' and g' w_(2^s).  btw[2^(t-1) + b] holds the constant of
    // block b of the t-th stage from the top (fr_coset_twiddle_kernel): g^(n / 2^t) w_(2^t)^rev(b).  Stages run from the top down as for `dif`; in a pass over
    // the high stages a tile sees a handful of blocks, so its twiddle loads are broadcasts instead of 32-byte pieces of as many cache lines.
    const uint64_t* btw;
};
__global__ void __launch_bounds__(256) fr_fft_pass_kernel(FrPass a) {
    __shared__ uint32_t lds[8][1024];
    const uint32_t tile_elems = min(1024u, 1u << a.log_n), C = tile_elems >> a.ns, log_c = 31 - __clz(C);
    const uint32_t tid = threadIdx.x;
    // tiles: (hi, c_blk) with c_blk < 2^s0 / C
    const uint32_t cblks = (1u << a.s0) >> log_c;
    const uint64_t hi = blockIdx.x / cblks, c0 = (uint64_t)(blockIdx.x % cblks) << log_c;
    const uint64_t base = (hi << (a.s0 + a.ns)) + c0;
    for (uint32_t e = tid; e < tile_elems; e += 256) {
        const uint32_t r = e >> log_c, c = e & (C - 1);
        const uint64_t i = base + ((uint64_t)r << a.s0) + c;
        u256 x;
        if (a.first) {
            const uint64_t src = a.no_gather ? i : __brevll(i) >> (64 - a.log_n);
            if (src < a.n_in) {
                x = a.in_mont ? load256(a.in + 4 * src) : m_from_int<F_R>(load256(a.in + 4 * src));
                if (a.pre) x = m_mul<F_R>(x, load256(a.pre + 4 * src));
            } else x = u_zero();
        } else x = load256(a.in + 4 * i);
#pragma unroll
        for (int l = 0; l < 8; l++) lds[l][e] = x.l[l];
    }
    __syncthreads();
    // (Two stages per LDS round trip -- four rows per lane in registers -- were built and measured: the pass kernel grows from 81 to 170 VGPRs
    // (three waves per SIMD instead of six) and evaluate_h at k = 23 went from 594 to 641 ms, 744 ms capped at 128 VGPRs with spills.  One
    // stage per round trip at six waves stays.  Round 4, after the product was written by hand: the product inlined into the butterfly (no call,
    // no operand moves) and both of a thread's twiddles fetched before its first product changed nothing -- evaluate_h 452.5 and 458 against
    // 451 ms -- the pass is neither call- nor twiddle-latency-bound; it runs at 0.72 of the VALU rate of its mix.  The butterflies on nine
    // 29-bit limbs (bn254_f29.cuh: inlined product against twiddles in the 2^261 form, sums reduced on the top limb, 36 KB of LDS) were built
    // too: evaluate_h 442 against 435 ms -- the product it saves is paid back in limb planes, normalisations and a block less per CU.)
    const bool block_tw = a.btw != nullptr, dif_fly = a.dif && !block_tw;
    for (uint32_t it = 1; it <= a.ns; it++) {
        const uint32_t st = a.dif ? a.ns + 1 - it : it;
        const uint32_t s = a.s0 + st, lh = st - 1, half = 1u << lh;
        for (uint32_t b = tid; b < tile_elems / 2; b += 256) {
            const uint32_t q = b >> log_c, c = b & (C - 1);
            const uint32_t pos = q & (half - 1);
            const uint32_t r_lo = ((q >> lh) << (lh + 1)) | pos;
            const uint32_t e0 = (r_lo << log_c) | c, e1 = e0 + (half << log_c);
            const uint64_t j = ((uint64_t)pos << a.s0) + c0 + c;
            // block form: the pair's block of 2^s values is number (global index >> s) = hi 2^(ns - st) + (q >> lh)
            const u256 w = block_tw ? load256(a.btw + 4 * ((1ull << (a.log_n - s)) + (hi << (a.ns - st)) + (q >> lh)))
                                    : load256(a.tw + 4 * (j << (a.log_n - s)));
            u256 u, v;
#pragma unroll
            for (int l = 0; l < 8; l++) { u.l[l] = lds[l][e0]; v.l[l] = lds[l][e1]; }
            u256 p, m;
            if (dif_fly) { p = m_add<F_R>(u, v); m = m_mul<F_R>(m_sub<F_R>(u, v), w); }
            else { v = m_mul<F_R>(v, w); p = m_add<F_R>(u, v); m = m_sub<F_R>(u, v); }
#pragma unroll
            for (int l = 0; l < 8; l++) { lds[l][e0] = p.l[l]; lds[l][e1] = m.l[l]; }
        }
        __syncthreads();
    }
    for (uint32_t e = tid; e < tile_elems; e += 256) {
        const uint32_t r = e >> log_c, c = e & (C - 1);
        const uint64_t i = base + ((uint64_t)r << a.s0) + c;
        u256 x;
#pragma unroll
        for (int l = 0; l < 8; l++) x.l[l] = lds[l][e];
        if (a.last) {
            if (i >= a.n_out) continue;
            if (a.out_mont) {
                if (a.post) x = m_mul<F_R>(x, load256(a.post + 4 * i));
                else if (a.use_scale) x = m_mul<F_R>(x, a.scale);
            } else if (a.post) x = m_canon<F_R>(m_mul<F_R>(x, load256(a.post + 4 * i)));
            else x = a.use_scale ? m_canon<F_R>(m_mul<F_R>(x, a.scale)) : m_to_int<F_R>(x);
        }
        store256(a.out + 4 * i, x);
    }
}

// ================================================================ G1 ================================================
struct jac { u256 x, y, z; };                 // z == 0 (mod q): the identity
GL_DEV jac j_identity() { jac p; p.x = u_const(BN254C_FQ_ONE); p.y = p.x; p.z = u_zero(); return p; }
GL_DEV bool j_is_identity(const jac& p) { return m_is_zero<F_Q>(p.z); }
__device__ __noinline__ jac j_double(jac p) {
    if (j_is_identity(p)) return p;
    const u256 a = m_mul<F_Q>(p.x, p.x), b = m_mul<F_Q>(p.y, p.y), c = m_mul<F_Q>(b, b);
    const u256 xb = m_add<F_Q>(p.x, b);
    u256 d = m_sub<F_Q>(m_sub<F_Q>(m_mul<F_Q>(xb, xb), a), c);
    d = m_add<F_Q>(d, d);
    const u256 e = m_add<F_Q>(m_add<F_Q>(a, a), a), f = m_mul<F_Q>(e, e);
    jac r;
    r.x = m_sub<F_Q>(f, m_add<F_Q>(d, d));
    u256 c8 = m_add<F_Q>(c, c);
    c8 = m_add<F_Q>(c8, c8);
    c8 = m_add<F_Q>(c8, c8);
    r.y = m_sub<F_Q>(m_mul<F_Q>(e, m_sub<F_Q>(d, r.x)), c8);
    const u256 yz = m_mul<F_Q>(p.y, p.z);
    r.z = m_add<F_Q>(yz, yz);
    return r;
}
// p + (x2, y2) with an affine second operand (Montgomery form; the caller skips the identity)
__device__ __noinline__ jac j_madd(jac p, u256 x2, u256 y2) {
    if (j_is_identity(p)) { jac r; r.x = x2; r.y = y2; r.z = u_const(BN254C_FQ_ONE); return r; }
    const u256 z1z1 = m_mul<F_Q>(p.z, p.z);
    const u256 u2 = m_mul<F_Q>(x2, z1z1), s2 = m_mul<F_Q>(m_mul<F_Q>(y2, p.z), z1z1);
    const u256 h = m_sub<F_Q>(u2, p.x), r = m_sub<F_Q>(s2, p.y);
    if (m_is_zero<F_Q>(h)) return m_is_zero<F_Q>(r) ? j_double(p) : j_identity();
    const u256 h2 = m_mul<F_Q>(h, h), h3 = m_mul<F_Q>(h2, h), v = m_mul<F_Q>(p.x, h2);
    jac o;
    o.x = m_sub<F_Q>(m_sub<F_Q>(m_mul<F_Q>(r, r), h3), m_add<F_Q>(v, v));
    o.y = m_sub<F_Q>(m_mul<F_Q>(r, m_sub<F_Q>(v, o.x)), m_mul<F_Q>(p.y, h3));
    o.z = m_mul<F_Q>(p.z, h);
    return o;
}
// the same, inlined into its caller: as a call the mixed addition makes every kernel that uses it a 210-VGPR kernel (two waves per SIMD: the
// convention keeps the callee's whole frame apart from the caller's); inlined, the bucket loops take 142 (three waves) -- the bucket
// accumulation is bound by the latency of its random 64-byte point reads, which more resident waves cover
GL_DEV jac j_madd_inl(const jac& p, const u256& x2, const u256& y2) {
    if (j_is_identity(p)) { jac r; r.x = x2; r.y = y2; r.z = u_const(BN254C_FQ_ONE); return r; }
    const u256 z1z1 = m_mul<F_Q>(p.z, p.z);
    const u256 u2 = m_mul<F_Q>(x2, z1z1), s2 = m_mul<F_Q>(m_mul<F_Q>(y2, p.z), z1z1);
    const u256 h = m_sub<F_Q>(u2, p.x), r = m_sub<F_Q>(s2, p.y);
    if (m_is_zero<F_Q>(h)) return m_is_zero<F_Q>(r) ? j_double(p) : j_identity();
    const u256 h2 = m_mul<F_Q>(h, h), h3 = m_mul<F_Q>(h2, h), v = m_mul<F_Q>(p.x, h2);
    jac o;
    o.x = m_sub<F_Q>(m_sub<F_Q>(m_mul<F_Q>(r, r), h3), m_add<F_Q>(v, v));
    o.y = m_sub<F_Q>(m_mul<F_Q>(r, m_sub<F_Q>(v, o.x)), m_mul<F_Q>(p.y, h3));
    o.z = m_mul<F_Q>(p.z, h);
    return o;
}
__device__ __noinline__ jac j_add(jac p, jac q) {
    if (j_is_identity(p)) return q;
    if (j_is_identity(q)) return p;
    const u256 z1z1 = m_mul<F_Q>(p.z, p.z), z2z2 = m_mul<F_Q>(q.z, q.z);
    const u256 u1 = m_mul<F_Q>(p.x, z2z2), u2 = m_mul<F_Q>(q.x, z1z1);
    const u256 s1 = m_mul<F_Q>(m_mul<F_Q>(p.y, q.z), z2z2), s2 = m_mul<F_Q>(m_mul<F_Q>(q.y, p.z), z1z1);
    const u256 h = m_sub<F_Q>(u2, u1), r = m_sub<F_Q>(s2, s1);
    if (m_is_zero<F_Q>(h)) return m_is_zero<F_Q>(r) ? j_double(p) : j_identity();
    const u256 h2 = m_mul<F_Q>(h, h), h3 = m_mul<F_Q>(h2, h), v = m_mul<F_Q>(u1, h2);
    jac o;
    o.x = m_sub<F_Q>(m_sub<F_Q>(m_mul<F_Q>(r, r), h3), m_add<F_Q>(v, v));
    o.y = m_sub<F_Q>(m_mul<F_Q>(r, m_sub<F_Q>(v, o.x)), m_mul<F_Q>(s1, h3));
    o.z = m_mul<F_Q>(m_mul<F_Q>(p.z, q.z), h);
    return o;
}
// inlined form for the reduction levels (same reason as j_madd_inl)
GL_DEV jac j_add_inl(const jac& p, const jac& q) {
    if (j_is_identity(p)) return q;
    if (j_is_identity(q)) return p;
    const u256 z1z1 = m_mul<F_Q>(p.z, p.z), z2z2 = m_mul<F_Q>(q.z, q.z);
    const u256 u1 = m_mul<F_Q>(p.x, z2z2), u2 = m_mul<F_Q>(q.x, z1z1);
    const u256 s1 = m_mul<F_Q>(m_mul<F_Q>(p.y, q.z), z2z2), s2 = m_mul<F_Q>(m_mul<F_Q>(q.y, p.z), z1z1);
    const u256 h = m_sub<F_Q>(u2, u1), r = m_sub<F_Q>(s2, s1);
    if (m_is_zero<F_Q>(h)) return m_is_zero<F_Q>(r) ? j_double(p) : j_identity();
    const u256 h2 = m_mul<F_Q>(h, h), h3 = m_mul<F_Q>(h2, h), v = m_mul<F_Q>(u1, h2);
    jac o;
    o.x = m_sub<F_Q>(m_sub<F_Q>(m_mul<F_Q>(r, r), h3), m_add<F_Q>(v, v));
    o.y = m_sub<F_Q>(m_mul<F_Q>(r, m_sub<F_Q>(v, o.x)), m_mul<F_Q>(s1, h3));
    o.z = m_mul<F_Q>(m_mul<F_Q>(p.z, q.z), h);
    return o;
}
GL_DEV void j_store(uint32_t* dst, const jac& p) {
#pragma unroll
    for (int j = 0; j < 8; j++) { dst[j] = p.x.l[j]; dst[8 + j] = p.y.l[j]; dst[16 + j] = p.z.l[j]; }
}
GL_DEV jac j_load(const uint32_t* src) {
    jac p;
#pragma unroll
    for (int j = 0; j < 8; j++) { p.x.l[j] = src[j]; p.y.l[j] = src[8 + j]; p.z.l[j] = src[16 + j]; }
    return p;
}

struct MsmArgs {
    const uint64_t* points;     // [n][8] affine x | y, canonical integers; (0, 0) = identity
    const uint64_t* scalars;    // [n][4]
    uint64_t n;
    uint32_t c, n_windows;      // window bits, windows (signed digits: |digit| <= 2^(c-1), one more window takes the last carry)
    uint32_t wps, n_sets;       // windows per scalar set, scalar sets sharing the bases (n_windows = wps * n_sets: a set's windows are just more windows)
    uint32_t cb;                // c - 1: a window has 2^cb buckets, bucket j collects the points whose digit is +-(j + 1)
    uint32_t* pm;               // [n][16] points in Montgomery form
    uint32_t* hist;             // [W][2^cb]     counts, then exclusive offsets
    uint32_t* cursor;           // [W][2^cb]     scatter cursors
    uint32_t* idx;              // [W][n]        point indices sorted by bucket, bit 31 = the digit is negative
    uint32_t* buckets;          // [W][2^cb][24] Jacobian bucket sums
    uint32_t* order;            // [W * 2^cb]    bucket ids (w << cb | j) by decreasing size
    uint32_t* size_hist;        // [MSM_SIZE_BINS] buckets per size, then the write cursor of each size class
    uint32_t* wsum;             // [W][24]       window sums
    uint64_t* result;           // [8]
    // buckets of more than MSM_BIG points (skewed scalars: selector columns of 0 / 1, constants, the sparse top window) are summed by
    // whole workgroups instead of one lane
    uint32_t* big_counters;     // [4]           work items, big buckets; of those, the items / buckets of more than MSM_MID points (workgroup path)
    uint32_t* big_items;        // [max_items][2] bucket id, first point of the item (relative to the bucket)
    uint32_t* big_buckets;      // [max_big][3]  bucket id, first item, items
    uint32_t* big_partial;      // [max_items][24]
    uint32_t max_items, max_big;
    // two-level sort (msm_digits_kernel ... msm_fine_sort_kernel): signed digits, (index | sign, low digit bits) pairs grouped by the
    // digit's high bits, and the per-window counters of those coarse bins
    uint32_t* dig;              // [W][n]     magnitude | sign << 31; 0 = nothing to add
    uint32_t* pairs;            // [W][n][2]
    uint32_t* coarse_cnt;       // [W][2^cbits] points per coarse bin
    uint32_t* coarse_start;     // [W][2^cbits] exclusive scan of the counts
    uint32_t* coarse_fill;      // [W][2^cbits] reservation cursors of the scatter
    uint32_t cbits, chunk;      // coarse bits (cb - MSM_FINE_BITS), points per block of the coarse kernels
    uint32_t have_table;        // pm is a prepared table (gl355_bn254_g1_msm_prepare): msm_digits_kernel converts nothing
    // regions of the fine sort with more than MSM_FINE_BIG pairs (skewed scalars: runs of equal values put a window's points into one coarse bin) are
    // cut into slices of MSM_FINE_SLICE pairs, one workgroup each (msm_fine_big_* kernels)
    uint32_t* fb_counters;      // [2]            big regions, slices
    uint32_t* fb_regions;       // [max_reg][2]   region (w << cbits | bin), slices
    uint32_t* fb_items;         // [max_items][2] region, slice
    uint32_t fb_max_reg, fb_max_items;
};
#define MSM_FINE_BITS 10u
#define MSM_FINE_BIG (1u << 17)   // pairs in a (window, coarse bin) region above which it is sorted by several workgroups
#define MSM_FINE_SLICE (1u << 15)
#define MSM_UNROLL 8
#define MSM_SIZE_BINS 128
#define MSM_BIG 256u            // a lane sums at most this many points; a normal bucket holds 8-64
#define MSM_BIG_WG_POINTS 4096u // points per workgroup item of a big bucket (16 per lane), more when a bucket would need over 256 items
GL_DEV uint32_t msm_digit(const uint64_t* k, uint32_t w, uint32_t c) {
    const uint32_t bit = w * c;
    if (bit >= 256) return 0;
    const uint32_t limb = bit >> 6, off = bit & 63;
    uint64_t v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (uint32_t)(v & ((1ull << c) - 1));
}
// signed digit of window w given the carry out of the windows below: magnitude (0 = nothing to add) and sign; raw digits of
// 2^(c-1) and more become negative and carry one into the next window, which halves the buckets of a window
GL_DEV uint32_t msm_signed_digit(const uint64_t* k, uint32_t w, uint32_t c, uint32_t& carry, bool& neg) {
    const uint32_t raw = msm_digit(k, w, c) + carry;
    neg = raw >= (1u << (c - 1));
    carry = neg ? 1u : 0u;
    return neg ? (1u << c) - raw : raw;
}
// counter[key] += 1 for every lane with `active`, returning the value before the lane's increment.  Lanes of a wave that share one of
// up to three sampled keys are combined into one atomic: with skewed scalars (all equal, 0 / 1, the sparse top window) half a million
// points hit ONE counter, and one-by-one atomics on it took 6 ms per pass; uniformly random keys cost three ballots more.
GL_DEV uint32_t msm_wave_inc(uint32_t* counter, uint32_t key, bool active) {
    uint32_t slot = 0;
    bool done = !active;
    uint64_t tried = 0;
    const uint32_t lane = __lane_id();
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const uint64_t cand = __ballot(!done) & ~tried;
        if (!cand) break;                                          // wave-uniform
        const int leader = __ffsll((unsigned long long)cand) - 1;
        tried |= 1ull << leader;
        const uint32_t lk = __shfl(key, leader);
        const uint64_t same = __ballot(!done && key == lk);
        const uint32_t cnt = (uint32_t)__popcll(same);
        if (cnt < 4) continue;                                     // not worth a round trip: leave them to the plain atomics
        uint32_t b = 0;
        if (lane == (uint32_t)leader) b = atomicAdd(counter + lk, cnt);
        b = __shfl(b, leader);
        if (!done && key == lk) { slot = b + (uint32_t)__popcll(same & ((1ull << lane) - 1)); done = true; }
    }
    if (!done) slot = atomicAdd(counter + key, 1u);
    return slot;
}
// GL355_MSM_F29 (default 1): the bucket loops add in the 29-bit-limb form of bn254_f29.cuh, whose Montgomery radix is 2^261: the point table
// then holds x 2^261, y 2^261 (one product by the Montgomery form of 32 more per coordinate and MSM).  0: the 8 x 32-bit mixed addition.
#ifndef GL355_MSM_F29
#define GL355_MSM_F29 1
#endif
GL_DEV u256 msm_table_form(const u256& v_mont) {
#if GL355_MSM_F29
    return m_canon<F_Q>(m_mul<F_Q>(v_mont, u_const(FQ_C32)));
#else
    return v_mont;
#endif
}
__global__ void msm_prepare_kernel(MsmArgs a) {          // points to Montgomery form + digit histograms
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    bool live = i < a.n;                                  // every lane stays for the wave-level combining below
    const uint64_t ii = live ? i : 0;
    const u256 x = load256(a.points + 8 * ii), y = load256(a.points + 8 * ii + 4);
    const bool ident = u_is_zero(x) && u_is_zero(y);
    if (live) {
        const u256 xm = ident ? u_zero() : msm_table_form(m_from_int<F_Q>(x)), ym = ident ? u_zero() : msm_table_form(m_from_int<F_Q>(y));
        uint32_t* d = a.pm + 16 * i;
#pragma unroll
        for (int j = 0; j < 8; j++) { d[j] = xm.l[j]; d[8 + j] = ym.l[j]; }
    }
    live = live && !ident;
    for (uint32_t set = 0; set < a.n_sets; set++) {
        const uint64_t* k = a.scalars + 4 * ((uint64_t)set * a.n + ii);
        uint32_t carry = 0;
        for (uint32_t w = 0; w < a.wps; w++) {
            bool neg;
            const uint32_t mag = msm_signed_digit(k, w, a.c, carry, neg);
            (void)msm_wave_inc(a.hist + ((uint64_t)(set * a.wps + w) << a.cb), mag ? mag - 1 : 0, live && mag != 0);
        }
    }
}
// per window: exclusive scan of the 2^cb counts (one workgroup), offsets copied to the cursors
__global__ void __launch_bounds__(1024) msm_scan_kernel(MsmArgs a) {
    __shared__ uint32_t sh[1024];
    const uint32_t w = blockIdx.x, nb = 1u << a.cb, tid = threadIdx.x;
    uint32_t* h = a.hist + ((uint64_t)w << a.cb);
    uint32_t* cur = a.cursor + ((uint64_t)w << a.cb);
    const uint32_t per = (nb + 1023) / 1024, lo = tid * per, hi = min(nb, lo + per);
    uint32_t s = 0;
    for (uint32_t b = lo; b < hi; b++) s += h[b];
    sh[tid] = s;
    __syncthreads();
    for (int st = 1; st < 1024; st <<= 1) {
        const uint32_t o = tid >= (uint32_t)st ? sh[tid - st] : 0;
        __syncthreads();
        sh[tid] += o;
        __syncthreads();
    }
    uint32_t run = tid ? sh[tid - 1] : 0;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t cnt = h[b];
        h[b] = run; cur[b] = run;
        run += cnt;
    }
}
__global__ void msm_scatter_kernel(MsmArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    bool live = i < a.n;
    const uint64_t ii = live ? i : 0;
    const uint32_t* d = a.pm + 16 * ii;
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) o |= d[j];
    live = live && o != 0;                                // the identity contributes nothing
    for (uint32_t set = 0; set < a.n_sets; set++) {
        const uint64_t* k = a.scalars + 4 * ((uint64_t)set * a.n + ii);
        uint32_t carry = 0;
        for (uint32_t ww = 0; ww < a.wps; ww++) {
            bool neg;
            const uint32_t mag = msm_signed_digit(k, ww, a.c, carry, neg), w = set * a.wps + ww;
            const bool act = live && mag != 0;
            const uint32_t pos = msm_wave_inc(a.cursor + ((uint64_t)w << a.cb), mag ? mag - 1 : 0, act);
            if (act) a.idx[(uint64_t)w * a.n + pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
        }
    }
}
// ---- prepared bases (round 6): tab[w][i] = 2^(c w) P_i in the bucket loops' table form, w < wps.  One lane per base walks its windows by c doublings
// each (Jacobian, 8 x 32-bit form), keeps X | Y in the table slot and Z and the running product of the Zs in scratch, inverts the product once
// and walks back (Montgomery's trick along the lane's own chain: 3 products per window instead of an inversion).  A multiple of a point of this
// prime-order group is never the identity and never has y = 0, so no Z is zero.
struct MsmTabArgs {
    const uint64_t* points;     // [n][8] affine, canonical
    uint32_t* tab;              // [wps][n][16]
    uint32_t* zs;               // [wps - 1][chunk][8]  Z of window w + 1 (Montgomery)
    uint32_t* pre;              // [wps - 1][chunk][8]  Z_1 ... Z_(w + 1)
    uint64_t n, i0, chunk;      // this launch covers bases [i0, min(n, i0 + chunk))
    uint32_t c, wps;
};
__global__ void __launch_bounds__(256) msm_table_build_kernel(MsmTabArgs a) {
    const uint64_t li = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, i = a.i0 + li;
    if (li >= a.chunk || i >= a.n) return;
    const u256 x = load256(a.points + 8 * i), y = load256(a.points + 8 * i + 4);
    if (u_is_zero(x) && u_is_zero(y)) {
        for (uint32_t w = 0; w < a.wps; w++) {
            uint32_t* t = a.tab + 16ull * ((uint64_t)w * a.n + i);
#pragma unroll
            for (int j = 0; j < 16; j++) t[j] = 0;
        }
        return;
    }
    jac q;
    q.x = m_from_int<F_Q>(x); q.y = m_from_int<F_Q>(y); q.z = u_const(BN254C_FQ_ONE);
    {
        const u256 xt = msm_table_form(q.x), yt = msm_table_form(q.y);
        uint32_t* t = a.tab + 16ull * i;
#pragma unroll
        for (int j = 0; j < 8; j++) { t[j] = xt.l[j]; t[8 + j] = yt.l[j]; }
    }
    u256 prefix = q.z;
    for (uint32_t w = 1; w < a.wps; w++) {
        for (uint32_t d = 0; d < a.c; d++) q = j_double(q);
        uint32_t* t = a.tab + 16ull * ((uint64_t)w * a.n + i);
        uint32_t* z = a.zs + 8ull * ((uint64_t)(w - 1) * a.chunk + li);
        uint32_t* pr = a.pre + 8ull * ((uint64_t)(w - 1) * a.chunk + li);
        prefix = w == 1 ? q.z : m_mul<F_Q>(prefix, q.z);
#pragma unroll
        for (int j = 0; j < 8; j++) { t[j] = q.x.l[j]; t[8 + j] = q.y.l[j]; z[j] = q.z.l[j]; pr[j] = prefix.l[j]; }
    }
    if (a.wps < 2) return;
    u256 inv = m_inv<F_Q>(prefix);                            // 1 / (Z_1 ... Z_(wps - 1))
    for (uint32_t w = a.wps - 1; w >= 1; w--) {
        uint32_t* t = a.tab + 16ull * ((uint64_t)w * a.n + i);
        const uint32_t* z = a.zs + 8ull * ((uint64_t)(w - 1) * a.chunk + li);
        u256 zi = inv, zw, X, Y;
        if (w > 1) {
            const uint32_t* pr = a.pre + 8ull * ((uint64_t)(w - 2) * a.chunk + li);
            u256 pw;
#pragma unroll
            for (int j = 0; j < 8; j++) pw.l[j] = pr[j];
            zi = m_mul<F_Q>(inv, pw);                         // 1 / Z_w
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { zw.l[j] = z[j]; X.l[j] = t[j]; Y.l[j] = t[8 + j]; }
        inv = m_mul<F_Q>(inv, zw);
        const u256 zi2 = m_mul<F_Q>(zi, zi);
        const u256 xt = msm_table_form(m_mul<F_Q>(X, zi2)), yt = msm_table_form(m_mul<F_Q>(Y, m_mul<F_Q>(zi2, zi)));
#pragma unroll
        for (int j = 0; j < 8; j++) { t[j] = xt.l[j]; t[8 + j] = yt.l[j]; }
    }
}
// ---- the sort in two levels (round 3).  The histogram and the scatter above pay one DEVICE-scope atomic per point and window each
// (2 x 109 M at 2^23 points: 4.0 + 6.0 of 36.6 ms -- the L2s of the eight XCDs are not coherent, so those atomics execute at the memory
// side).  Here the digits are written once (msm_digits_kernel), blocks of `chunk` points count and scatter them by part of the digit's bits
// with LDS atomics and one global atomic per (block, coarse bin), and one workgroup per (window, coarse bin) sorts its region by the other
// MSM_FINE_BITS bits in LDS, writing the bucket offsets (hist / cursor) and the final index array.  The coarse bin is the digit's LOW bits:
// the top window of a 254-bit scalar has only a few significant bits, and binned by the high bits its 2^23 points fell into 8 regions of
// a million points each, one workgroup per region (5 ms).  Same outputs as msm_prepare /
// msm_scan / msm_scatter up to the order of the points inside a bucket, which does not matter.  Skewed scalars only make regions large
// (a workgroup then loops over its region); nothing overflows.
__global__ void msm_digits_kernel(MsmArgs a) {            // points to Montgomery form + the signed digits of every window
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    bool ident;
    if (a.have_table) {                                       // the table's first slice IS pm; the identity is all zeros there too
        const uint4* t = reinterpret_cast<const uint4*>(a.pm + 16 * i);
        const uint4 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
        ident = (t0.x | t0.y | t0.z | t0.w | t1.x | t1.y | t1.z | t1.w | t2.x | t2.y | t2.z | t2.w | t3.x | t3.y | t3.z | t3.w) == 0;
    } else {
        const u256 x = load256(a.points + 8 * i), y = load256(a.points + 8 * i + 4);
        ident = u_is_zero(x) && u_is_zero(y);
        const u256 xm = ident ? u_zero() : msm_table_form(m_from_int<F_Q>(x)), ym = ident ? u_zero() : msm_table_form(m_from_int<F_Q>(y));
        uint32_t* d = a.pm + 16 * i;
#pragma unroll
        for (int j = 0; j < 8; j++) { d[j] = xm.l[j]; d[8 + j] = ym.l[j]; }
    }
    for (uint32_t set = 0; set < a.n_sets; set++) {
        const uint64_t* k = a.scalars + 4 * ((uint64_t)set * a.n + i);
        uint32_t carry = 0;
        for (uint32_t w = 0; w < a.wps; w++) {
            bool neg;
            const uint32_t mag = msm_signed_digit(k, w, a.c, carry, neg);
            a.dig[(uint64_t)(set * a.wps + w) * a.n + i] = ident ? 0u : (mag | (neg && mag ? 0x80000000u : 0u));
        }
    }
}
__global__ void __launch_bounds__(256) msm_coarse_count_kernel(MsmArgs a) {
    extern __shared__ uint32_t lh[];
    const uint32_t w = blockIdx.y, nbin = 1u << a.cbits;
    const uint64_t lo = (uint64_t)blockIdx.x * a.chunk, hi = min(a.n, lo + a.chunk);
    for (uint32_t b = threadIdx.x; b < nbin; b += 256) lh[b] = 0;
    __syncthreads();
    const uint32_t* dg = a.dig + (uint64_t)w * a.n;
    // (MSM_UNROLL loads in flight per thread: with one load per iteration these loops were chains of 64 dependent memory round trips)
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * MSM_UNROLL) {
        uint32_t d[MSM_UNROLL];
#pragma unroll
        for (int k = 0; k < MSM_UNROLL; k++) { const uint64_t i = i0 + 256ull * k; d[k] = i < hi ? dg[i] : 0u; }
#pragma unroll
        for (int k = 0; k < MSM_UNROLL; k++) {
            const uint32_t mag = d[k] & 0x7fffffffu;
            if (mag) atomicAdd(&lh[(mag - 1) & (nbin - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbin; b += 256)
        if (lh[b]) atomicAdd(a.coarse_cnt + (uint64_t)w * nbin + b, lh[b]);
}
__global__ void __launch_bounds__(1024) msm_coarse_scan_kernel(MsmArgs a) {     // per window: exclusive scan of the coarse counts
    __shared__ uint32_t sh[1024];
    const uint32_t w = blockIdx.x, nbin = 1u << a.cbits, tid = threadIdx.x;
    const uint32_t* c = a.coarse_cnt + (uint64_t)w * nbin;
    uint32_t* st = a.coarse_start + (uint64_t)w * nbin;
    const uint32_t per = (nbin + 1023) / 1024, lo = min(nbin, tid * per), hi = min(nbin, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += c[b];
    sh[tid] = sum;
    __syncthreads();
    for (int stp = 1; stp < 1024; stp <<= 1) {
        const uint32_t o = tid >= (uint32_t)stp ? sh[tid - stp] : 0;
        __syncthreads();
        sh[tid] += o;
        __syncthreads();
    }
    uint32_t run = tid ? sh[tid - 1] : 0;
    for (uint32_t b = lo; b < hi; b++) { st[b] = run; run += c[b]; }
}
__global__ void __launch_bounds__(256) msm_coarse_scatter_kernel(MsmArgs a) {
    extern __shared__ uint32_t lh[];                       // [2^cbits] counts, then running cursors; [2^cbits] bases
    const uint32_t w = blockIdx.y, nbin = 1u << a.cbits;
    uint32_t* lbase = lh + nbin;
    const uint64_t lo = (uint64_t)blockIdx.x * a.chunk, hi = min(a.n, lo + a.chunk);
    for (uint32_t b = threadIdx.x; b < nbin; b += 256) lh[b] = 0;
    __syncthreads();
    const uint32_t* dg = a.dig + (uint64_t)w * a.n;
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * MSM_UNROLL) {
        uint32_t d[MSM_UNROLL];
#pragma unroll
        for (int k = 0; k < MSM_UNROLL; k++) { const uint64_t i = i0 + 256ull * k; d[k] = i < hi ? dg[i] : 0u; }
#pragma unroll
        for (int k = 0; k < MSM_UNROLL; k++) {
            const uint32_t mag = d[k] & 0x7fffffffu;
            if (mag) atomicAdd(&lh[(mag - 1) & (nbin - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbin; b += 256) {
        const uint32_t cnt = lh[b];
        lbase[b] = cnt ? a.coarse_start[(uint64_t)w * nbin + b] + atomicAdd(a.coarse_fill + (uint64_t)w * nbin + b, cnt) : 0;
        lh[b] = 0;
    }
    __syncthreads();
    uint32_t* pr = a.pairs + 2ull * (uint64_t)w * a.n;
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * MSM_UNROLL) {
        uint32_t dd[MSM_UNROLL];
#pragma unroll
        for (int k = 0; k < MSM_UNROLL; k++) { const uint64_t i = i0 + 256ull * k; dd[k] = i < hi ? dg[i] : 0u; }
#pragma unroll
        for (int k = 0; k < MSM_UNROLL; k++) {
            const uint32_t d = dd[k], mag = d & 0x7fffffffu;
            if (!mag) continue;
            const uint32_t bin = (mag - 1) & (nbin - 1);
            const uint32_t pos = lbase[bin] + atomicAdd(&lh[bin], 1u);
            *reinterpret_cast<uint2*>(pr + 2ull * pos) = make_uint2((uint32_t)(i0 + 256ull * k) | (d & 0x80000000u), (mag - 1) >> a.cbits);
        }
    }
}
__global__ void __launch_bounds__(256) msm_fine_sort_kernel(MsmArgs a) {
    __shared__ uint32_t fh[1u << MSM_FINE_BITS], fbase[1u << MSM_FINE_BITS], part[256];
    const uint32_t bin = blockIdx.x, w = blockIdx.y, nbin = 1u << a.cbits, tid = threadIdx.x;
    const uint32_t start = a.coarse_start[(uint64_t)w * nbin + bin], cnt = a.coarse_cnt[(uint64_t)w * nbin + bin];
    constexpr uint32_t NF = 1u << MSM_FINE_BITS, PER = NF / 256;
    if (cnt > MSM_FINE_BIG) return;                                   // msm_fine_big_* (one workgroup would walk the whole region alone)
    for (uint32_t f = tid; f < NF; f += 256) fh[f] = 0;
    __syncthreads();
    const uint32_t* pr = a.pairs + 2ull * ((uint64_t)w * a.n + start);
    for (uint32_t k0 = tid; k0 < cnt; k0 += 256 * MSM_UNROLL) {
        uint32_t f[MSM_UNROLL];
#pragma unroll
        for (int j = 0; j < MSM_UNROLL; j++) { const uint32_t k = k0 + 256u * j; f[j] = k < cnt ? pr[2ull * k + 1] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < MSM_UNROLL; j++) if (f[j] != 0xFFFFFFFFu) atomicAdd(&fh[f[j]], 1u);
    }
    __syncthreads();
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) { c[j] = fh[tid * PER + j]; sum += c[j]; }
    part[tid] = sum;
    __syncthreads();
    for (int stp = 1; stp < 256; stp <<= 1) {
        const uint32_t o = tid >= (uint32_t)stp ? part[tid - stp] : 0;
        __syncthreads();
        part[tid] += o;
        __syncthreads();
    }
    uint32_t run = start + (tid ? part[tid - 1] : 0);
    const uint64_t bucket0 = ((uint64_t)w << a.cb) + bin;            // bucket = digit - 1 = fine << cbits | bin
#pragma unroll
    for (uint32_t j = 0; j < PER; j++) {
        const uint32_t f = tid * PER + j;
        a.hist[bucket0 + ((uint64_t)f << a.cbits)] = run; a.cursor[bucket0 + ((uint64_t)f << a.cbits)] = run + c[j];
        fbase[f] = run; fh[f] = 0;
        run += c[j];
    }
    __syncthreads();
    uint32_t* out = a.idx + (uint64_t)w * a.n;
    for (uint32_t k0 = tid; k0 < cnt; k0 += 256 * MSM_UNROLL) {
        uint2 pp[MSM_UNROLL];
#pragma unroll
        for (int j = 0; j < MSM_UNROLL; j++) { const uint32_t k = k0 + 256u * j; pp[j] = k < cnt ? *reinterpret_cast<const uint2*>(pr + 2ull * k) : make_uint2(0u, 0xFFFFFFFFu); }
#pragma unroll
        for (int j = 0; j < MSM_UNROLL; j++) if (pp[j].y != 0xFFFFFFFFu) out[fbase[pp[j].y] + atomicAdd(&fh[pp[j].y], 1u)] = pp[j].x;
    }
}
// ---- big regions of the fine sort.  A column whose values come in long runs (a grand product that stands still where its constraint is switched off, a
// constant, 0 / 1 flags) puts millions of pairs into ONE (window, coarse bin) region, and the single workgroup of msm_fine_sort_kernel walked it alone:
// 15 - 28 ms per call in a k = 23 proof against 1 - 7 for uniform columns.  Such regions are listed and cut into slices; a slice's workgroup counts its fine
// bins in LDS and adds them to the bucket counts (a.hist, zeroed by the caller), one workgroup per region turns counts into offsets, and the slices reserve
// their ranges per fine bin with one atomic each and scatter.  Same outputs as msm_fine_sort_kernel (hist = start, cursor = end of every bucket, idx).
__global__ void __launch_bounds__(256) msm_fine_big_list_kernel(MsmArgs a) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (a.n_windows << a.cbits)) return;
    const uint32_t cnt = a.coarse_cnt[r];
    if (cnt <= MSM_FINE_BIG) return;
    const uint32_t slices = (cnt + MSM_FINE_SLICE - 1) / MSM_FINE_SLICE;
    const uint32_t slot = atomicAdd(a.fb_counters, 1u), first = atomicAdd(a.fb_counters + 1, slices);
    if (slot >= a.fb_max_reg || first + slices > a.fb_max_items) return;      // cannot happen: the bounds are sums over all pairs (host side)
    a.fb_regions[2 * slot] = r; a.fb_regions[2 * slot + 1] = slices;
    for (uint32_t k = 0; k < slices; k++) { a.fb_items[2 * (first + k)] = r; a.fb_items[2 * (first + k) + 1] = k; }
}
// the fine-bin histogram of pairs [lo, hi) of a region in LDS (fh zeroed by the caller)
GL_DEV void msm_fine_slice_hist(const uint32_t* pr, uint32_t lo, uint32_t hi, uint32_t* fh) {
    for (uint32_t k0 = lo + threadIdx.x; k0 < hi; k0 += 256 * MSM_UNROLL) {
        uint32_t f[MSM_UNROLL];
#pragma unroll
        for (int j = 0; j < MSM_UNROLL; j++) { const uint32_t k = k0 + 256u * j; f[j] = k < hi ? pr[2ull * k + 1] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < MSM_UNROLL; j++) if (f[j] != 0xFFFFFFFFu) atomicAdd(&fh[f[j]], 1u);
    }
}
__global__ void __launch_bounds__(256) msm_fine_big_count_kernel(MsmArgs a) {
    __shared__ uint32_t fh[1u << MSM_FINE_BITS];
    constexpr uint32_t NF = 1u << MSM_FINE_BITS;
    const uint32_t n_items = min(a.fb_counters[1], a.fb_max_items), nbin = 1u << a.cbits;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t r = a.fb_items[2 * it], sl = a.fb_items[2 * it + 1], w = r >> a.cbits, bin = r & (nbin - 1);
        const uint32_t start = a.coarse_start[r], cnt = a.coarse_cnt[r];
        const uint32_t lo = sl * MSM_FINE_SLICE, hi = min(cnt, lo + MSM_FINE_SLICE);
        for (uint32_t f = threadIdx.x; f < NF; f += 256) fh[f] = 0;
        __syncthreads();
        msm_fine_slice_hist(a.pairs + 2ull * ((uint64_t)w * a.n + start), lo, hi, fh);
        __syncthreads();
        const uint64_t bucket0 = ((uint64_t)w << a.cb) + bin;
        for (uint32_t f = threadIdx.x; f < NF; f += 256) if (fh[f]) atomicAdd(a.hist + bucket0 + ((uint64_t)f << a.cbits), fh[f]);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) msm_fine_big_scan_kernel(MsmArgs a) {      // counts -> offsets, one workgroup per big region
    __shared__ uint32_t part[256];
    constexpr uint32_t NF = 1u << MSM_FINE_BITS, PER = NF / 256;
    const uint32_t n_reg = min(a.fb_counters[0], a.fb_max_reg), nbin = 1u << a.cbits, tid = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < n_reg; q += gridDim.x) {
        const uint32_t r = a.fb_regions[2 * q], w = r >> a.cbits, bin = r & (nbin - 1);
        const uint64_t bucket0 = ((uint64_t)w << a.cb) + bin;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) { c[j] = a.hist[bucket0 + ((uint64_t)(tid * PER + j) << a.cbits)]; sum += c[j]; }
        part[tid] = sum;
        __syncthreads();
        for (int stp = 1; stp < 256; stp <<= 1) {
            const uint32_t o = tid >= (uint32_t)stp ? part[tid - stp] : 0;
            __syncthreads();
            part[tid] += o;
            __syncthreads();
        }
        uint32_t run = a.coarse_start[r] + (tid ? part[tid - 1] : 0);
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint64_t b = bucket0 + ((uint64_t)(tid * PER + j) << a.cbits);
            a.hist[b] = run; a.cursor[b] = run;                       // the scatter's reservations move the cursor to the bucket's end
            run += c[j];
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) msm_fine_big_scatter_kernel(MsmArgs a) {
    __shared__ uint32_t fh[1u << MSM_FINE_BITS], fbase[1u << MSM_FINE_BITS];
    constexpr uint32_t NF = 1u << MSM_FINE_BITS;
    const uint32_t n_items = min(a.fb_counters[1], a.fb_max_items), nbin = 1u << a.cbits;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t r = a.fb_items[2 * it], sl = a.fb_items[2 * it + 1], w = r >> a.cbits, bin = r & (nbin - 1);
        const uint32_t start = a.coarse_start[r], cnt = a.coarse_cnt[r];
        const uint32_t lo = sl * MSM_FINE_SLICE, hi = min(cnt, lo + MSM_FINE_SLICE);
        const uint32_t* pr = a.pairs + 2ull * ((uint64_t)w * a.n + start);
        for (uint32_t f = threadIdx.x; f < NF; f += 256) fh[f] = 0;
        __syncthreads();
        msm_fine_slice_hist(pr, lo, hi, fh);
        __syncthreads();
        const uint64_t bucket0 = ((uint64_t)w << a.cb) + bin;
        for (uint32_t f = threadIdx.x; f < NF; f += 256) {
            fbase[f] = fh[f] ? atomicAdd(a.cursor + bucket0 + ((uint64_t)f << a.cbits), fh[f]) : 0;
            fh[f] = 0;
        }
        __syncthreads();
        uint32_t* out = a.idx + (uint64_t)w * a.n;
        for (uint32_t k0 = lo + threadIdx.x; k0 < hi; k0 += 256 * MSM_UNROLL) {
            uint2 pp[MSM_UNROLL];
#pragma unroll
            for (int j = 0; j < MSM_UNROLL; j++) { const uint32_t k = k0 + 256u * j; pp[j] = k < hi ? *reinterpret_cast<const uint2*>(pr + 2ull * k) : make_uint2(0u, 0xFFFFFFFFu); }
#pragma unroll
            for (int j = 0; j < MSM_UNROLL; j++) if (pp[j].y != 0xFFFFFFFFu) out[fbase[pp[j].y] + atomicAdd(&fh[pp[j].y], 1u)] = pp[j].x;
        }
        __syncthreads();
    }
}
// Buckets by decreasing size.  A lane sums one bucket, so a wave takes as long as its largest bucket: with 2^20 points in 2^16
// buckets per window the sizes are Poisson(16) and the largest of 64 is ~27 -- 40 % of the lanes' time idle.  Sizes are small
// integers, so a counting sort (per-workgroup LDS histogram, one global atomic per class and workgroup) puts equal sizes side by side.
GL_DEV uint32_t msm_bucket_size(const MsmArgs& a, uint32_t id) {
    const uint32_t sz = a.cursor[id] - a.hist[id];                 // cursor = end of the bucket's range after the scatter
    return sz < MSM_SIZE_BINS ? sz : MSM_SIZE_BINS - 1;
}
__global__ void __launch_bounds__(256) msm_size_hist_kernel(MsmArgs a) {
    __shared__ uint32_t h[MSM_SIZE_BINS];
    if (threadIdx.x < MSM_SIZE_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x, total = a.n_windows << a.cb;
    if (id < total) atomicAdd(&h[msm_bucket_size(a, id)], 1u);
    __syncthreads();
    if (threadIdx.x < MSM_SIZE_BINS && h[threadIdx.x]) atomicAdd(a.size_hist + threadIdx.x, h[threadIdx.x]);
}
__global__ void msm_size_scan_kernel(MsmArgs a) {                   // counts -> start of each size class, largest size first
    if (threadIdx.x || blockIdx.x) return;
    uint32_t run = 0;
    for (int sz = MSM_SIZE_BINS - 1; sz >= 0; sz--) { const uint32_t cnt = a.size_hist[sz]; a.size_hist[sz] = run; run += cnt; }
}
__global__ void __launch_bounds__(256) msm_order_kernel(MsmArgs a) {
    __shared__ uint32_t h[MSM_SIZE_BINS], base[MSM_SIZE_BINS];
    if (threadIdx.x < MSM_SIZE_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x, total = a.n_windows << a.cb;
    uint32_t sz = 0, slot = 0;
    if (id < total) { sz = msm_bucket_size(a, id); slot = atomicAdd(&h[sz], 1u); }
    __syncthreads();
    if (threadIdx.x < MSM_SIZE_BINS && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(a.size_hist + threadIdx.x, h[threadIdx.x]);
    __syncthreads();
    if (id < total) a.order[base[sz] + slot] = id;
}
GL_DEV jac msm_add_point(const MsmArgs& a, jac acc, uint32_t e) {      // e: point index, bit 31 = subtract
    const uint32_t* p = a.pm + 16ull * (e & 0x7fffffffu);
    u256 x, y;
#pragma unroll
    for (int j = 0; j < 8; j++) { x.l[j] = p[j]; y.l[j] = p[8 + j]; }
    if (e >> 31) y = m_sub<F_Q>(u_zero(), y);
    return j_madd_inl(acc, x, y);
}
#if GL355_MSM_F29
// The accumulator of a bucket loop in the 29-bit-limb form.  Bounds (in units of q; "n" = limbs normalised): x < 5.2 n, y < 3.3 n, z < 1.3 n.
struct jac29 { f29 x, y, z; bool ident; };
GL_DEV jac jac29_lower(const jac29& p) {                            // -> the R = 2^256 Jacobian form of the reduction kernels
    if (p.ident) return j_identity();
    jac r;
    r.x = f29_lower(p.x); r.y = f29_lower(p.y); r.z = f29_lower(p.z);
    return r;
}
// the rare branch: the bucket's sum and the point share their x.  Equal points: the sum is twice the AFFINE point (a = 0 curve:
// XX = x^2, YY = y^2, S = 2 ((x + YY)^2 - XX - YY^2), M = 3 XX, X3 = M^2 - 2 S, Y3 = M (S - X3) - 8 YY^2, Z3 = 2 y), in the same lazy form with
// a product by one wherever a bound would pass what the lent constants cover; opposite points: the identity.
GL_DEV void jac29_same_x(jac29& acc, const f29& x2, const f29& y2, bool equal) {
    if (!equal) { acc.ident = true; return; }
    const f29 one = f29_const(FQ29_ONE);
    const f29 xx = f29_mul(x2, x2), yy = f29_mul(y2, y2), yyyy = f29_mul(yy, yy);
    const f29 t = f29_norm(f29_add(x2, yy));
    const f29 s0 = f29_norm(f29_sub(f29_mul(t, t), f29_norm(f29_add(xx, yyyy)), FQ29_C4));          // < 5.1
    const f29 sv = f29_mul(f29_norm(f29_add(s0, s0)), one);                                             // S < 1.1
    const f29 mv = f29_norm(f29_add(xx, f29_add(xx, xx)));                                              // M < 3.1
    const f29 x3 = f29_norm(f29_sub(f29_mul(mv, mv), f29_norm(f29_add(sv, sv)), FQ29_C4));            // < 5.1
    const f29 y4 = f29_norm(f29_add(f29_norm(f29_add(yyyy, yyyy)), f29_norm(f29_add(yyyy, yyyy))));    // 4 YY^2 < 4.2
    f29 y3 = f29_mul(f29_sub(sv, x3, FQ29_C8), mv);
    y3 = f29_norm(f29_sub(y3, y4, FQ29_C8));
    y3 = f29_norm(f29_sub(y3, y4, FQ29_C8));                                                            // < 17.3
    acc.x = x3; acc.y = f29_mul(y3, one); acc.z = f29_mul(f29_norm(f29_add(y2, y2)), one);
}
// acc += (x2, +-y2): 11 products, sums and differences without carries, five re-normalisations
GL_DEV void msm_add_point29(const MsmArgs& a, jac29& acc, uint32_t e) {
    const uint32_t* p = a.pm + 16ull * (e & 0x7fffffffu);
    u256 x8, y8;
#pragma unroll
    for (int j = 0; j < 8; j++) { x8.l[j] = p[j]; y8.l[j] = p[8 + j]; }
    const f29 x2 = f29_from_u256(x8);
    f29 y2 = f29_from_u256(y8);                                      // < q, n
    if (e >> 31) y2 = f29_norm(f29_neg(y2, FQ29_C2));                 // 2 q - y < 2 q, n
    if (acc.ident) { acc.x = x2; acc.y = y2; acc.z = f29_const(FQ29_ONE); acc.ident = false; return; }
    const f29 z1z1 = f29_mul(acc.z, acc.z);
    const f29 u2 = f29_mul(x2, z1z1), s2 = f29_mul(f29_mul(y2, acc.z), z1z1);
    const f29 h = f29_norm(f29_sub(u2, acc.x, FQ29_C8));            // < 9.3, n
    const f29 h2 = f29_mul(h, h);
    const f29 r = f29_norm(f29_sub(s2, acc.y, FQ29_C4));            // < 5.3, n
    if (f29_is_zero_mod(h2)) { jac29_same_x(acc, x2, y2, f29_is_zero_mod(f29_mul(r, r))); return; }
    const f29 h3 = f29_mul(h2, h), v = f29_mul(acc.x, h2);
    const f29 w = f29_norm(f29_add(h3, f29_add(v, v)));              // h^3 + 2 v < 3.9, n
    const f29 x3 = f29_norm(f29_sub(f29_mul(r, r), w, FQ29_C4));     // < 5.3, n
    const f29 m1 = f29_mul(f29_sub(v, x3, FQ29_C8), r);              // (v + 8 q - x3 < 9.3) r
    const f29 y3 = f29_norm(f29_sub(m1, f29_mul(acc.y, h3), FQ29_C2));   // < 3.3, n
    acc.z = f29_mul(acc.z, h);
    acc.x = x3; acc.y = y3;
}
// ---- Jacobian + Jacobian and doubling in the same form, for the reduction levels (chains of dependent additions on few lanes: the regime
// where the inlined 29-bit product is twice the called asm one).  Coordinates stay below 12 q with normalised limbs: inputs (X, Y, Z) < 12 q
// give X3 < 5.2, Y3 < 3.3, Z3 < 1.1 (addition) and X3 < 9.3, Y3 < 1.2, Z3 < 2.2 (doubling).
GL_DEV void jac29_double(jac29& p) {
    if (p.ident) return;
    const f29 one = f29_const(FQ29_ONE);
    const f29 A = f29_mul(p.x, p.x), B = f29_mul(p.y, p.y), C = f29_mul(B, B);
    const f29 t = f29_norm(f29_add(p.x, B));
    const f29 d0 = f29_norm(f29_sub(f29_mul(t, t), f29_norm(f29_add(A, C)), FQ29_C4));        // (X + B)^2 - A - C < 6.2
    const f29 dr = f29_mul(d0, one);
    const f29 D = f29_norm(f29_add(dr, dr));                                                     // < 2.1
    const f29 E = f29_norm(f29_add(A, f29_add(A, A)));                                           // < 5.6
    const f29 x3 = f29_norm(f29_sub(f29_mul(E, E), f29_norm(f29_add(D, D)), FQ29_C8));          // < 9.3
    const f29 c4 = f29_norm(f29_add(f29_norm(f29_add(C, C)), f29_norm(f29_add(C, C))));          // 4 C < 4.1
    f29 y3 = f29_mul(f29_sub(D, x3, FQ29_C16), E);                                               // (D + 16 q - X3 < 18.2) E
    y3 = f29_norm(f29_sub(y3, c4, FQ29_C8));
    y3 = f29_norm(f29_sub(y3, c4, FQ29_C8));                                                     // < 17.7
    const f29 yz = f29_mul(p.y, p.z);
    p.x = x3; p.y = f29_mul(y3, one); p.z = f29_norm(f29_add(yz, yz));
}
GL_DEV void jac29_add(jac29& p, const jac29& q) {
    if (q.ident) return;
    if (p.ident) { p = q; return; }
    const f29 z1z1 = f29_mul(p.z, p.z), z2z2 = f29_mul(q.z, q.z);
    const f29 u1 = f29_mul(p.x, z2z2), u2 = f29_mul(q.x, z1z1);
    const f29 s1 = f29_mul(p.y, f29_mul(q.z, z2z2)), s2 = f29_mul(q.y, f29_mul(p.z, z1z1));
    const f29 h = f29_norm(f29_sub(u2, u1, FQ29_C2)), r = f29_norm(f29_sub(s2, s1, FQ29_C2));
    const f29 h2 = f29_mul(h, h);
    if (f29_is_zero_mod(h2)) {                                       // the same x: twice the point, or the identity
        if (f29_is_zero_mod(f29_mul(r, r))) jac29_double(p); else p.ident = true;
        return;
    }
    const f29 h3 = f29_mul(h2, h), v = f29_mul(u1, h2);
    const f29 w = f29_norm(f29_add(h3, f29_add(v, v)));
    const f29 x3 = f29_norm(f29_sub(f29_mul(r, r), w, FQ29_C4));
    const f29 m1 = f29_mul(f29_sub(v, x3, FQ29_C8), r);
    p.y = f29_norm(f29_sub(m1, f29_mul(s1, h3), FQ29_C2));
    p.z = f29_mul(f29_mul(p.z, q.z), h);
    p.x = x3;
}
GL_DEV jac29 jac29_lift(const jac& p) {                              // the 8 x 32-bit R-domain point in this form (three products)
    jac29 r;
    r.ident = j_is_identity(p);
    if (!r.ident) { r.x = f29_lift_inl(p.x); r.y = f29_lift_inl(p.y); r.z = f29_lift_inl(p.z); }
    return r;
}
// ---- the bucket loops' accumulator in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2): the mixed addition is 10 products where the Jacobian one is
// 11 -- ZZ and ZZZ are kept instead of being rebuilt from Z (z^2, y2 z) at every step.  madd-2008-s: U2 = x2 ZZ, S2 = y2 ZZZ, P = U2 - X, R = S2 - Y,
// PP = P^2, PPP = P PP, Q = X PP, X3 = R^2 - PPP - 2 Q, Y3 = R (Q - X3) - Y PPP, ZZ3 = ZZ PP, ZZZ3 = ZZZ PPP; the sums and differences carry the bounds of the
// Jacobian form above (P < 9.3, R < 5.3, X3 < 5.3, Y3 < 3.3; ZZ, ZZZ are products: < 1.3).  A finished sum leaves as the Jacobian point (X ZZ, Y ZZZ, ZZ).
struct xyzz29 { f29 x, y, zz, zzz; bool ident; };
GL_DEV jac msm_xyzz_lower(const xyzz29& p) {
    if (p.ident) return j_identity();
    jac r;
    r.x = f29_lower(f29_mul(p.x, p.zz)); r.y = f29_lower(f29_mul(p.y, p.zzz)); r.z = f29_lower(p.zz);
    return r;
}
GL_DEV void msm_add_point_xyzz(const MsmArgs& a, xyzz29& acc, uint32_t e) {
    const uint32_t* p = a.pm + 16ull * (e & 0x7fffffffu);
    u256 x8, y8;
#pragma unroll
    for (int j = 0; j < 8; j++) { x8.l[j] = p[j]; y8.l[j] = p[8 + j]; }
    const f29 x2 = f29_from_u256(x8);
    f29 y2 = f29_from_u256(y8);                                      // < q, n
    if (e >> 31) y2 = f29_norm(f29_neg(y2, FQ29_C2));                 // 2 q - y < 2 q, n
    if (acc.ident) { acc.x = x2; acc.y = y2; acc.zz = f29_const(FQ29_ONE); acc.zzz = acc.zz; acc.ident = false; return; }
    const f29 u2 = f29_mul(x2, acc.zz), s2 = f29_mul(y2, acc.zzz);
    const f29 h = f29_norm(f29_sub(u2, acc.x, FQ29_C8));            // P < 9.3, n
    const f29 h2 = f29_mul(h, h);
    const f29 r = f29_norm(f29_sub(s2, acc.y, FQ29_C4));            // R < 5.3, n
    if (f29_is_zero_mod(h2)) {                                       // the same x: twice the affine point, or the identity (rare)
        jac29 t;
        t.ident = false;
        jac29_same_x(t, x2, y2, f29_is_zero_mod(f29_mul(r, r)));
        if (t.ident) { acc.ident = true; return; }
        acc.x = t.x; acc.y = t.y; acc.zz = f29_mul(t.z, t.z); acc.zzz = f29_mul(acc.zz, t.z);
        return;
    }
    const f29 h3 = f29_mul(h2, h), v = f29_mul(acc.x, h2);
    const f29 w = f29_norm(f29_add(h3, f29_add(v, v)));              // PPP + 2 Q < 3.9, n
    const f29 x3 = f29_norm(f29_sub(f29_mul(r, r), w, FQ29_C4));     // < 5.3, n
    const f29 m1 = f29_mul(f29_sub(v, x3, FQ29_C8), r);              // (Q + 8 q - X3 < 9.3) R
    const f29 y3 = f29_norm(f29_sub(m1, f29_mul(acc.y, h3), FQ29_C2));   // < 3.3, n
    acc.zz = f29_mul(acc.zz, h2);
    acc.zzz = f29_mul(acc.zzz, h3);
    acc.x = x3; acc.y = y3;
}
#ifndef GL355_MSM_XYZZ
#define GL355_MSM_XYZZ 1
#endif
#if GL355_MSM_XYZZ
#define MSM_ACC_T xyzz29
#define MSM_ACC_INIT(A) xyzz29 A; A.ident = true
#define MSM_ACC_ADD(ARGS, A, E) msm_add_point_xyzz(ARGS, A, E)
#define MSM_ACC_JAC(A) msm_xyzz_lower(A)
#else
#define MSM_ACC_T jac29
#define MSM_ACC_INIT(A) jac29 A; A.ident = true
#define MSM_ACC_ADD(ARGS, A, E) msm_add_point29(ARGS, A, E)
#define MSM_ACC_JAC(A) jac29_lower(A)
#endif
#else
#define MSM_ACC_T jac
#define MSM_ACC_INIT(A) jac A = j_identity()
#define MSM_ACC_ADD(ARGS, A, E) A = msm_add_point(ARGS, A, E)
#define MSM_ACC_JAC(A) (A)
#endif
// one lane per (window, bucket), taken in the order above: sum of the bucket's points (big buckets: the kernels below)
__global__ void __launch_bounds__(256) msm_bucket_kernel(MsmArgs a) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (a.n_windows << a.cb)) return;
    const uint32_t id = a.order[g], w = id >> a.cb;
    const uint32_t lo = a.hist[id], hi = a.cursor[id];
    if (hi - lo > MSM_BIG) return;
    MSM_ACC_INIT(acc);
    for (uint32_t k = lo; k < hi; k++) MSM_ACC_ADD(a, acc, a.idx[(uint64_t)w * a.n + k]);
    j_store(a.buckets + (uint64_t)id * 24, MSM_ACC_JAC(acc));
}
// ---- big buckets: a work list (bucket, slice) built on the device, one workgroup per slice (lanes stride through the slice, then a
// tree over the 256 lane sums in LDS), one workgroup per big bucket for the slices' sums.  One lane per bucket made a 2^20-point MSM
// whose scalars were all equal -- or whose top window held one digit -- a matter of seconds (2^19 dependent additions).
// Buckets of MSM_BIG < sz <= MSM_MID points (the top window of a uniform 2^23-point MSM: 2^13 buckets of ~1024) are cut into items of
// MSM_MID_SLICE points summed by ONE LANE each, like ordinary buckets, and their <= 32 partial sums are added by one lane per bucket: a
// workgroup per 1024-point bucket spent its time in the 8-level tree over 256 lane sums of 4 points each (4.0 of 38.5 ms at 2^23).
#define MSM_MID 8192u
#define MSM_MID_SLICE 64u
GL_DEV uint32_t msm_big_slice(uint32_t sz) {                       // points per item: at most 256 items per bucket
    // (mid-size buckets: at most 32 lane items of 64 .. 256 points -- with prepared bases and 22-bit windows the 12 significant bits of the top window make
    // 2^11 buckets of 2^12 points each, which took the workgroup path at 2.4 times the lane path's cost per addition while MSM_MID was 2048)
    if (sz <= MSM_MID) return max(MSM_MID_SLICE, (sz + 31) / 32);
    const uint32_t need = (sz + 255) / 256;
    return need > MSM_BIG_WG_POINTS ? ((need + 255) & ~255u) : MSM_BIG_WG_POINTS;
}
__global__ void __launch_bounds__(256) msm_big_list_kernel(MsmArgs a) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (a.n_windows << a.cb)) return;
    const uint32_t sz = a.cursor[id] - a.hist[id];
    if (sz <= MSM_BIG) return;
    const uint32_t slice = msm_big_slice(sz), cnt = (sz + slice - 1) / slice;
    const uint32_t first = atomicAdd(a.big_counters, cnt), slot = atomicAdd(a.big_counters + 1, 1u);
    if (sz > MSM_MID) { atomicAdd(a.big_counters + 2, cnt); atomicAdd(a.big_counters + 3, 1u); }
    if (first + cnt > a.max_items || slot >= a.max_big) return;   // cannot happen: the bounds are sums over all points (host side)
    for (uint32_t c = 0; c < cnt; c++) { a.big_items[2 * (first + c)] = id; a.big_items[2 * (first + c) + 1] = c * slice; }
    a.big_buckets[3 * slot] = id; a.big_buckets[3 * slot + 1] = first; a.big_buckets[3 * slot + 2] = cnt;
}
GL_DEV jac msm_wg_tree(jac acc, uint32_t* sh /* 256 x 24 */) {     // sum of the 256 lanes' points, valid on lane 0
    const uint32_t t = threadIdx.x;
    j_store(sh + 24 * t, acc);
    __syncthreads();
    for (uint32_t st = 128; st >= 1; st >>= 1) {
        if (t < st) { acc = j_add(acc, j_load(sh + 24 * (t + st))); j_store(sh + 24 * t, acc); }
        __syncthreads();
    }
    return acc;
}
// items of mid-size buckets: one lane per item (a fixed grid strides over the work list)
__global__ void __launch_bounds__(256) msm_mid_partial_kernel(MsmArgs a) {
    const uint32_t n_items = *a.big_counters;
    for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += gridDim.x * blockDim.x) {
        const uint32_t id = a.big_items[2 * it], off = a.big_items[2 * it + 1], w = id >> a.cb;
        const uint32_t base = a.hist[id], end = a.cursor[id];
        if (end - base > MSM_MID) continue;                        // a workgroup item (below)
        const uint32_t lo = base + off, hi = min(end, lo + msm_big_slice(end - base));
        MSM_ACC_INIT(acc);
        for (uint32_t k = lo; k < hi; k++) MSM_ACC_ADD(a, acc, a.idx[(uint64_t)w * a.n + k]);
        j_store(a.big_partial + 24ull * it, MSM_ACC_JAC(acc));
    }
}
// ... and one lane per mid-size bucket for its partial sums
__global__ void __launch_bounds__(64) msm_mid_final_kernel(MsmArgs a) {
    const uint32_t n_big = a.big_counters[1];
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_big; b += gridDim.x * blockDim.x) {
        const uint32_t id = a.big_buckets[3 * b], first = a.big_buckets[3 * b + 1], cnt = a.big_buckets[3 * b + 2];
        if (a.cursor[id] - a.hist[id] > MSM_MID) continue;
        jac acc = j_load(a.big_partial + 24ull * first);
        for (uint32_t k = 1; k < cnt; k++) acc = j_add(acc, j_load(a.big_partial + 24ull * (first + k)));
        j_store(a.buckets + (uint64_t)id * 24, acc);
    }
}
__global__ void __launch_bounds__(256) msm_big_partial_kernel(MsmArgs a) {      // a fixed grid walks the work list (usually empty)
    __shared__ uint32_t sh[256 * 24];
    // (the list holds the lane items of the mid-size buckets too -- 2^17 of them for the top window of a uniform 2^23-point MSM -- and walking
    // it just to skip them was 2 ms per call: nothing to do unless some bucket takes the workgroup path)
    if (a.big_counters[2] == 0) return;
    const uint32_t n_items = *a.big_counters;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t id = a.big_items[2 * it], off = a.big_items[2 * it + 1], w = id >> a.cb;
        if (a.cursor[id] - a.hist[id] <= MSM_MID) continue;        // a lane item (above); block-uniform
        const uint32_t lo = a.hist[id] + off, end = a.cursor[id];
        const uint32_t hi = min(end, lo + msm_big_slice(end - a.hist[id]));
        MSM_ACC_INIT(acc29);
        for (uint32_t k = lo + threadIdx.x; k < hi; k += 256) MSM_ACC_ADD(a, acc29, a.idx[(uint64_t)w * a.n + k]);
        jac acc = MSM_ACC_JAC(acc29);
        acc = msm_wg_tree(acc, sh);
        if (threadIdx.x == 0) j_store(a.big_partial + 24ull * it, acc);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) msm_big_final_kernel(MsmArgs a) {
    __shared__ uint32_t sh[256 * 24];
    if (a.big_counters[3] == 0) return;
    const uint32_t n_big = a.big_counters[1];
    for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
        const uint32_t id = a.big_buckets[3 * b], first = a.big_buckets[3 * b + 1], cnt = a.big_buckets[3 * b + 2];
        if (a.cursor[id] - a.hist[id] <= MSM_MID) continue;        // msm_mid_final_kernel's; block-uniform
        jac acc = threadIdx.x < cnt ? j_load(a.big_partial + 24ull * (first + threadIdx.x)) : j_identity();
        acc = msm_wg_tree(acc, sh);
        if (threadIdx.x == 0) j_store(a.buckets + (uint64_t)id * 24, acc);
        __syncthreads();
    }
}

// Window sum  sum_j (j + 1) * B_j = Wt + S  by a recursion on pairs (S, Wt) = (sum of the items, sum of local index * item) over groups of 2^kbits
// items: a group of buckets gives S = sum B_b and Wt = sum (b - b0) B_b by the running-sum trick (S_run += B_b from the top,
// L += S_run); a group of pairs at the next level gives  S' = sum S_u,  Wt' = sum Wt_u + width * sum (u - u0) S_u  with
// width = the number of buckets one item spans (a power of two: `shift` doublings).  Every level is one launch of (windows x groups)
// lanes whose dependent chain is ~3 * 2^kbits additions; the single-lane tails of the first version (256 + 48 and then 256 dependent
// additions per window: 19 of 34 ms at 2^20 points) are gone.  The last level's single group gives the window sum Wt + S.
struct MsmLevel {
    const uint32_t* in_s;       // [W][t_in][24]
    const uint32_t* in_w;       // [W][t_in][24] or null (level 0: the items are the buckets themselves)
    uint32_t* out_s;            // [W][t_in >> kbits][24]
    uint32_t* out_w;
    uint32_t t_in, kbits, shift, n_windows;
};
__global__ void __launch_bounds__(64) msm_level_kernel(MsmLevel l) {
    const uint32_t groups = l.t_in >> l.kbits;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups * l.n_windows) return;
    const uint32_t w = g / groups, v = g % groups, k = 1u << l.kbits;
    const uint32_t* s_in = l.in_s + ((uint64_t)w * l.t_in + (uint64_t)v * k) * 24;
#if GL355_MSM_F29
    jac29 run, acc;
    run.ident = acc.ident = true;
    for (uint32_t u = k; u-- > 0;) {
        jac29_add(run, jac29_lift(j_load(s_in + u * 24)));
        if (u) jac29_add(acc, run);                                   // acc = sum_u u * S_u: item u is counted in the u sums taken at u' = u .. 1
    }
    for (uint32_t d = 0; d < l.shift; d++) jac29_double(acc);
    if (l.in_w) {
        const uint32_t* w_in = l.in_w + ((uint64_t)w * l.t_in + (uint64_t)v * k) * 24;
        for (uint32_t u = 0; u < k; u++) jac29_add(acc, jac29_lift(j_load(w_in + u * 24)));
    }
    j_store(l.out_s + ((uint64_t)w * groups + v) * 24, jac29_lower(run));
    j_store(l.out_w + ((uint64_t)w * groups + v) * 24, jac29_lower(acc));
#else
    jac run = j_identity(), acc = j_identity();
    for (uint32_t u = k; u-- > 0;) {
        run = j_add_inl(run, j_load(s_in + u * 24));
        if (u) acc = j_add_inl(acc, run);                             // acc = sum_u u * S_u: item u is counted in the u sums taken at u' = u .. 1
    }
    for (uint32_t d = 0; d < l.shift; d++) acc = j_double(acc);
    if (l.in_w) {
        const uint32_t* w_in = l.in_w + ((uint64_t)w * l.t_in + (uint64_t)v * k) * 24;
        for (uint32_t u = 0; u < k; u++) acc = j_add_inl(acc, j_load(w_in + u * 24));
    }
    j_store(l.out_s + ((uint64_t)w * groups + v) * 24, run);
    j_store(l.out_w + ((uint64_t)w * groups + v) * 24, acc);
#endif
}

// The same level with EIGHT LANES PER GROUP (kbits == 3).  A level is pure latency -- one lane's chain of ~26 dependent additions of ~30 us
// each, whatever the level's size: 7 levels were 5.7 of 38.5 ms at 2^23 points -- so the chain is cut instead: suffix sums of the eight S_u
// by a three-step scan across the lanes, sum_{u >= 1} run_u and sum_u Wt_u by three-step trees: 10 dependent additions + the doublings.
GL_DEV jac j_shfl_down(const jac& p, uint32_t d) {
    jac r;
#pragma unroll
    for (int j = 0; j < 8; j++) { r.x.l[j] = __shfl_down(p.x.l[j], d, 8); r.y.l[j] = __shfl_down(p.y.l[j], d, 8); r.z.l[j] = __shfl_down(p.z.l[j], d, 8); }
    return r;
}
#if GL355_MSM_F29
GL_DEV jac29 jac29_shfl_down(const jac29& p, uint32_t d) {
    jac29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) { r.x.l[j] = __shfl_down(p.x.l[j], d, 8); r.y.l[j] = __shfl_down(p.y.l[j], d, 8); r.z.l[j] = __shfl_down(p.z.l[j], d, 8); }
    r.ident = __shfl_down((int)p.ident, d, 8) != 0;
    return r;
}
__global__ void __launch_bounds__(64) msm_level_coop_kernel(MsmLevel l) {
    const uint32_t groups = l.t_in >> 3, total = groups * l.n_windows;
    const uint32_t gq = blockIdx.x * 8 + (threadIdx.x >> 3), u = threadIdx.x & 7;
    const bool live = gq < total;
    const uint32_t g = live ? gq : total - 1;                        // idle groups of the last block redo the last one (lanes stay for the shuffles)
    const uint32_t w = g / groups, v = g % groups;
    const uint64_t item = (uint64_t)w * l.t_in + (uint64_t)v * 8 + u;
    jac29 x = jac29_lift(j_load(l.in_s + item * 24));
#pragma unroll 1
    for (uint32_t d = 1; d < 8; d <<= 1) {                           // x_u = S_u + ... + S_7
        const jac29 y = jac29_shfl_down(x, d);
        if (u + d < 8) jac29_add(x, y);
    }
    jac29 acc = x;                                                    // sum_{u >= 1} run_u = sum_u u S_u
    if (!u) acc.ident = true;
#pragma unroll 1
    for (uint32_t d = 4; d >= 1; d >>= 1) {
        const jac29 y = jac29_shfl_down(acc, d);
        if (u < d) jac29_add(acc, y);
    }
    if (u == 0) for (uint32_t d = 0; d < l.shift; d++) jac29_double(acc);
    if (l.in_w) {
        jac29 wt = jac29_lift(j_load(l.in_w + item * 24));
#pragma unroll 1
        for (uint32_t d = 4; d >= 1; d >>= 1) {
            const jac29 y = jac29_shfl_down(wt, d);
            if (u < d) jac29_add(wt, y);
        }
        if (u == 0) jac29_add(acc, wt);
    }
    if (u == 0 && live) {
        j_store(l.out_s + ((uint64_t)w * groups + v) * 24, jac29_lower(x));
        j_store(l.out_w + ((uint64_t)w * groups + v) * 24, jac29_lower(acc));
    }
}
#else
__global__ void __launch_bounds__(64) msm_level_coop_kernel(MsmLevel l) {
    const uint32_t groups = l.t_in >> 3, total = groups * l.n_windows;
    const uint32_t gq = blockIdx.x * 8 + (threadIdx.x >> 3), u = threadIdx.x & 7;
    const bool live = gq < total;
    const uint32_t g = live ? gq : total - 1;                        // idle groups of the last block redo the last one (lanes stay for the shuffles)
    const uint32_t w = g / groups, v = g % groups;
    const uint64_t item = (uint64_t)w * l.t_in + (uint64_t)v * 8 + u;
    jac x = j_load(l.in_s + item * 24);
#pragma unroll 1
    for (uint32_t d = 1; d < 8; d <<= 1) {                           // x_u = S_u + ... + S_7
        const jac y = j_shfl_down(x, d);
        if (u + d < 8) x = j_add(x, y);
    }
    jac acc = u ? x : j_identity();                                  // sum_{u >= 1} run_u = sum_u u S_u
#pragma unroll 1
    for (uint32_t d = 4; d >= 1; d >>= 1) {
        const jac y = j_shfl_down(acc, d);
        if (u < d) acc = j_add(acc, y);
    }
    if (u == 0) for (uint32_t d = 0; d < l.shift; d++) acc = j_double(acc);
    if (l.in_w) {
        jac wt = j_load(l.in_w + item * 24);
#pragma unroll 1
        for (uint32_t d = 4; d >= 1; d >>= 1) {
            const jac y = j_shfl_down(wt, d);
            if (u < d) wt = j_add(wt, y);
        }
        if (u == 0) acc = j_add(acc, wt);
    }
    if (u == 0 && live) {
        j_store(l.out_s + ((uint64_t)w * groups + v) * 24, x);
        j_store(l.out_w + ((uint64_t)w * groups + v) * 24, acc);
    }
}
#endif

// ================================================================ fixed-base batch multiplication ===================
// out[i] = scalars[i] * base for one base point: what ParamsKZG::setup does for the powers of tau ([s^i] G, verifier_api.rs:77).
// 8-bit windows over a table T[w][d] = d * 2^(8 w) * base (32 x 256 affine points, built per call): one mixed addition per non-zero
// byte of the scalar and one inversion per output -- no doublings in the main loop.
struct FbArgs {
    const uint64_t* base;       // [8] affine, canonical integers
    const uint64_t* scalars;    // [n][4]
    uint64_t n;
    uint32_t* win;              // [32][24]       2^(8 w) * base, Jacobian
    uint32_t* table;            // [32][256][16]  affine Montgomery x | y; d = 0 unused
    uint64_t* out;              // [n][8]
};
GL_DEV void j_to_affine_mont(const jac& p, u256& x, u256& y) {           // p not the identity
    const u256 zi = m_inv<F_Q>(p.z), zi2 = m_mul<F_Q>(zi, zi);
    x = m_mul<F_Q>(p.x, zi2);
    y = m_mul<F_Q>(p.y, m_mul<F_Q>(zi2, zi));
}
__global__ void fb_windows_kernel(FbArgs a) {                            // lane w: 8 w doublings of the base
    const uint32_t w = threadIdx.x;
    if (w >= 32) return;
    const u256 x = load256(a.base), y = load256(a.base + 4);
    jac p;
    if (u_is_zero(x) && u_is_zero(y)) p = j_identity();
    else { p.x = m_from_int<F_Q>(x); p.y = m_from_int<F_Q>(y); p.z = u_const(BN254C_FQ_ONE); }
    for (uint32_t k = 0; k < 8 * w; k++) p = j_double(p);
    j_store(a.win + 24 * w, p);
}
__global__ void __launch_bounds__(256) fb_table_kernel(FbArgs a) {       // lane (w, d): d * B_w by double-and-add, to affine
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 32 * 256) return;
    const uint32_t d = g & 255;
    const jac b = j_load(a.win + 24 * (g >> 8));
    jac acc = j_identity();
    for (int bit = 7; bit >= 0; bit--) {
        acc = j_double(acc);
        if ((d >> bit) & 1) acc = j_add(acc, b);
    }
    u256 x = u_zero(), y = u_zero();
    if (!j_is_identity(acc)) j_to_affine_mont(acc, x, y);
    uint32_t* t = a.table + 16ull * g;
#pragma unroll
    for (int j = 0; j < 8; j++) { t[j] = x.l[j]; t[8 + j] = y.l[j]; }
}
__global__ void __launch_bounds__(256) fb_mul_kernel(FbArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t* k = a.scalars + 4 * i;
    jac acc = j_identity();
    for (uint32_t w = 0; w < 32; w++) {
        const uint32_t d = (uint32_t)(k[w >> 3] >> (8 * (w & 7))) & 255u;
        if (!d) continue;
        const uint32_t* t = a.table + 16ull * (w * 256 + d);
        u256 x, y;
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { x.l[j] = t[j]; y.l[j] = t[8 + j]; o |= t[j] | t[8 + j]; }
        if (o) acc = j_madd(acc, x, y);                                // a zero entry: d * B_w is the identity (base of small order)
    }
    uint64_t* dst = a.out + 8 * i;
    if (j_is_identity(acc)) {
#pragma unroll
        for (int j = 0; j < 8; j++) dst[j] = 0;
        return;
    }
    u256 x, y;
    j_to_affine_mont(acc, x, y);
    store256(dst, m_to_int<F_Q>(x));
    store256(dst + 4, m_to_int<F_Q>(y));
}

}  // namespace gl355

using namespace gl355;

// ================================================================ KZG composites ====================
// ParamsKZG::setup / commit / commit_lagrange and the single-point opening the SHPLONK prover reduces to, as the reference reaches them
// through verify_inside_snark (src/plonky2_verifier/verifier_api.rs:77-92, chip/native_chip/test_utils.rs:57-95; k = 23 in README.md:171-177).
// Scalars cross the ABI as plain 256-bit integers (4 x u64, any value: reduced on load); kernels work in Montgomery form.

// out[i] = tau^i (plain integers), i < n: the scalars of the powers-of-tau loop
__global__ void kzg_tau_powers_kernel(u256 tau_mont, uint64_t n, uint64_t* out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store256(out + 4 * i, m_to_int<F_R>(m_pow_u64<F_R>(tau_mont, i)));
}
// out[i] = L_i(tau) = (tau^n - 1) / n * w^i / (tau - w^i) (plain integers): the Lagrange basis of the 2^k domain at tau.  A lane takes KZG_LG_CHUNK
// consecutive i: one inversion per chunk (Montgomery's trick).  *bad is set if tau lies in the domain.
constexpr int KZG_LG_CHUNK = 16;
__global__ void __launch_bounds__(64) kzg_lagrange_kernel(u256 tau_mont, u256 w_mont, u256 w_inv_mont, u256 c_mont /* (tau^n - 1) / n */, uint64_t n,
                                                          uint64_t* out, uint64_t* pre /* n x 4 scratch */, uint32_t* bad) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t i0 = t * KZG_LG_CHUNK;
    if (i0 >= n) return;
    const uint64_t i1 = min(n, i0 + KZG_LG_CHUNK);
    u256 wi = m_pow_u64<F_R>(w_mont, i0);
    u256 acc = u_const(f_one<F_R>());
#pragma unroll 1
    for (uint64_t i = i0; i < i1; i++) {                 // forward: denominators (kept in out[]) and their running products
        const u256 d = m_sub<F_R>(tau_mont, wi);
        if (m_is_zero<F_R>(d)) atomicOr(bad, 1u);
        store256(pre + 4 * i, acc);
        store256(out + 4 * i, d);
        acc = m_mul<F_R>(acc, d);
        if (i + 1 < i1) wi = m_mul<F_R>(wi, w_mont);
    }
    u256 inv = m_inv<F_R>(acc);
#pragma unroll 1
    for (uint64_t i = i1; i-- > i0;) {                   // backward: 1 / d_i = inv * pre_i, then inv *= d_i; w^i steps down with w^-1
        const u256 d = load256(out + 4 * i);
        const u256 dinv = m_mul<F_R>(inv, load256(pre + 4 * i));
        inv = m_mul<F_R>(inv, d);
        store256(out + 4 * i, m_to_int<F_R>(m_mul<F_R>(m_mul<F_R>(c_mont, wi), dinv)));
        wi = m_mul<F_R>(wi, w_inv_mont);
    }
}
// Synthetic division by (X - z) as a blocked suffix Horner scan.  For an array A of m field elements and a point Z:
//     Q[i] = sum_{j > i} A[j] Z^(j - i - 1)  (i < m; Q[m-1] = 0),      E = sum_j A[j] Z^j.
// With A = the coefficients of p and Z = z: Q[0 .. n-2] are the coefficients of (p - p(z)) / (X - z) and E = p(z).
// Level kernels: (1) a lane's chunk value H_t = sum_{j in chunk t} A[j] Z^(j - start_t); the carries C_t = Q_H[t] of the array H at the
// point Z^chunk come from the next level (same problem, m / chunk elements); (3) a lane walks its chunk downwards from its carry.
constexpr uint32_t KZG_DIV_CHUNK = 64;
__global__ void kzg_div_chunk_kernel(const uint64_t* A, uint64_t m, u256 z_mont, int a_is_mont, uint64_t* H /* Montgomery */) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t s0 = t * KZG_DIV_CHUNK;
    if (s0 >= m) return;
    const uint64_t e0 = min(m, s0 + KZG_DIV_CHUNK);
    u256 h = u_zero();
#pragma unroll 1
    for (uint64_t j = e0; j-- > s0;) {
        const u256 a = a_is_mont ? load256(A + 4 * j) : m_from_int<F_R>(load256(A + 4 * j));
        h = m_add<F_R>(m_mul<F_R>(h, z_mont), a);
    }
    store256(H + 4 * t, h);
}
// carry == nullptr: the whole array is one chunk (m <= KZG_DIV_CHUNK), lane 0 only.  q_plain: write Q as plain integers (the top level)
__global__ void kzg_div_walk_kernel(const uint64_t* A, uint64_t m, u256 z_mont, int a_is_mont, const uint64_t* carry /* Montgomery, per chunk */,
                                    uint64_t* Q, int q_plain, uint64_t* E /* Montgomery; written by the lane of chunk 0 */) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t s0 = t * KZG_DIV_CHUNK;
    if (s0 >= m) return;
    const uint64_t e0 = min(m, s0 + KZG_DIV_CHUNK);
    u256 sacc = carry ? load256(carry + 4 * t) : u_zero();
#pragma unroll 1
    for (uint64_t j = e0; j-- > s0;) {
        store256(Q + 4 * j, q_plain ? m_to_int<F_R>(sacc) : sacc);         // Q[j] = the running suffix value before A[j] enters
        const u256 a = a_is_mont ? load256(A + 4 * j) : m_from_int<F_R>(load256(A + 4 * j));
        sacc = m_add<F_R>(m_mul<F_R>(sacc, z_mont), a);
    }
    if (t == 0 && E) store256(E, sacc);
}
__global__ void kzg_from_mont1_kernel(const uint64_t* in, uint64_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) store256(out, m_to_int<F_R>(load256(in)));
}

extern "C" {

}  // extern "C"

namespace gl355 {
// w_n^i (Montgomery), i < n / 2, for the forward or the inverse transform of 2^log_n points: what every pass of fr_fft_pass_kernel reads.
// `tw` holds n / 2 + 1 elements; scratch for the two seed tables comes from the context.
int32_t bn254_fr_twiddles(Ctx* ctx, uint32_t log_n, bool inverse, uint64_t* tw) {
    const uint64_t n = 1ull << log_n;
    H256 w = inverse ? H256{{BN254C_FR_ROOT_INV_64[0], BN254C_FR_ROOT_INV_64[1], BN254C_FR_ROOT_INV_64[2], BN254C_FR_ROOT_INV_64[3]}}
                     : H256{{BN254C_FR_ROOT_64[0], BN254C_FR_ROOT_64[1], BN254C_FR_ROOT_64[2], BN254C_FR_ROOT_64[3]}};
    for (uint32_t k = log_n; k < BN254C_FR_S; k++) w = h_mulmod(w, w);
    const uint64_t count = std::max<uint64_t>(1, n / 2), n_hi = (count + 1023) / 1024;
    Scratch seed(ctx);
    GL355_TRY(seed.get((1024 + n_hi) * 32));
    uint64_t* lo = seed.as<uint64_t>();
    uint64_t* hi = lo + 4 * 1024;
    hipLaunchKernelGGL(fr_twiddle_seed_kernel, dim3((uint32_t)((1024 + n_hi + 255) / 256)), dim3(256), 0, ctx->stream, lo, hi, n_hi, h_to_mont(w));
    hipLaunchKernelGGL(fr_twiddle_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, tw, count, lo, hi);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
// tab[i] = f * base^i (Montgomery), i < count (base, f: plain integers; f = 1 for plain powers)
int32_t bn254_fr_power_table(Ctx* ctx, const uint64_t base[4], const uint64_t f[4], uint64_t count, uint64_t* tab) {
    const uint64_t n_hi = (count + 1023) / 1024;
    Scratch seed(ctx);
    GL355_TRY(seed.get((1024 + n_hi) * 32));
    uint64_t* lo = seed.as<uint64_t>();
    uint64_t* hi = lo + 4 * 1024;
    hipLaunchKernelGGL(fr_twiddle_seed_kernel, dim3((uint32_t)((1024 + n_hi + 255) / 256)), dim3(256), 0, ctx->stream, lo, hi, n_hi, h_to_mont(h_from_words(base)));
    // (lo hi) R * (f R) * R^-1 = lo hi f R: Montgomery again
    hipLaunchKernelGGL(fr_power_mont_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, tab, count, lo, hi, h_to_mont(h_from_words(f)));
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
// The transform on resident data in Montgomery form: out[k] = post[k] * sum_i pre[i] in[i] w^(ik) (i < n_in, zero beyond; k < n_out), w from
// `tw` (bn254_fr_twiddles: the caller picks the direction and owns the 1 / n, e.g. inside `post` or as `scale`).  `work`: n elements of
// scratch; in may equal out.
// Decimation in frequency on resident Montgomery data: natural-order input (n_in values, zero beyond, times pre[i] if given), BIT-REVERSED
// output: out[bitrev(k)] = sum_i pre[i] in[i] w^(ik).  No pass gathers; the first one goes in -> out, the others run in place on `out`.
int32_t bn254_fr_ntt_mont_dif(Ctx* ctx, const uint64_t* in, uint64_t n_in, uint64_t* out, uint32_t log_n, const uint64_t* tw, const uint64_t* pre) {
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) {
        GL355_HIP(ctx, hipMemcpyAsync(out, in, 32, hipMemcpyDeviceToDevice, ctx->stream));
        return GL355_OK;
    }
    std::vector<uint32_t> ns;
    ns.push_back(std::min(10u, log_n));
    const uint32_t rem = log_n - ns[0], more = (rem + 6) / 7;
    for (uint32_t k = 0; k < more; k++) ns.push_back(rem / more + (k < rem % more ? 1 : 0));
    std::vector<uint32_t> s0s(ns.size());
    for (size_t k = 0, s0 = 0; k < ns.size(); k++) { s0s[k] = (uint32_t)s0; s0 += ns[k]; }
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, n / 1024);
    for (size_t k = ns.size(); k-- > 0;) {
        FrPass pa;
        memset(&pa, 0, sizeof pa);
        pa.first = k + 1 == ns.size(); pa.last = k == 0;
        pa.in = pa.first ? in : out;
        pa.out = out;
        pa.tw = tw; pa.log_n = log_n; pa.s0 = s0s[k]; pa.ns = ns[k];
        pa.in_mont = 1; pa.out_mont = 1; pa.no_gather = 1; pa.dif = 1;
        pa.n_in = n_in; pa.n_out = n;
        pa.pre = pre;
        hipLaunchKernelGGL(fr_fft_pass_kernel, dim3(tiles), dim3(256), 0, ctx->stream, pa);
    }
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
// The coset transform in block form: out[bitrev(k)] = sum_i in[i] shift^i w^(ik) for i < n_in (zero beyond), the same values bn254_fr_ntt_mont_dif gives with
// pre[i] = shift^i -- without the power table, its product per element and the scattered twiddle loads of the high stages (FrPass::btw).  `btw`: n elements,
// filled here for this shift (one product per entry: 0.1 ms at 2^23) -- callers transform many columns per shift and pass fill = false after the first.
int32_t bn254_fr_ntt_mont_coset_dif(Ctx* ctx, const uint64_t* in, uint64_t n_in, uint64_t* out, uint32_t log_n, const uint64_t* tw, const uint64_t shift_plain[4],
                                    uint64_t* btw, bool fill) {
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) {
        GL355_HIP(ctx, hipMemcpyAsync(out, in, 32, hipMemcpyDeviceToDevice, ctx->stream));
        return GL355_OK;
    }
    if (fill) {
        // gpow[t] = shift^(n / 2^t), t = 1 .. log_n (Montgomery): repeated squaring from the bottom
        std::vector<u256> gp(log_n + 1);
        memset(gp.data(), 0, gp.size() * sizeof(u256));
        H256 g = h_from_words(shift_plain);
        for (uint32_t t = log_n; t >= 1; t--) {
            gp[t] = h_to_mont(g);
            g = h_mulmod(g, g);
        }
        Scratch d(ctx);
        GL355_TRY(d.get(gp.size() * 32));
        GL355_HIP(ctx, hipMemcpyAsync(d.p, gp.data(), gp.size() * 32, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(fr_coset_twiddle_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, tw, d.as<uint64_t>(), log_n, btw);
        GL355_HIP(ctx, hipGetLastError());
        GL355_HIP(ctx, ctx->wait());                                     // gp is pageable host memory
    }
    std::vector<uint32_t> ns;
    ns.push_back(std::min(10u, log_n));
    const uint32_t rem = log_n - ns[0], more = (rem + 6) / 7;
    for (uint32_t k = 0; k < more; k++) ns.push_back(rem / more + (k < rem % more ? 1 : 0));
    std::vector<uint32_t> s0s(ns.size());
    for (size_t k = 0, s0 = 0; k < ns.size(); k++) { s0s[k] = (uint32_t)s0; s0 += ns[k]; }
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, n / 1024);
    for (size_t k = ns.size(); k-- > 0;) {
        FrPass pa;
        memset(&pa, 0, sizeof pa);
        pa.first = k + 1 == ns.size(); pa.last = k == 0;
        pa.in = pa.first ? in : out;
        pa.out = out;
        pa.tw = tw; pa.btw = btw; pa.log_n = log_n; pa.s0 = s0s[k]; pa.ns = ns[k];
        pa.in_mont = 1; pa.out_mont = 1; pa.no_gather = 1; pa.dif = 1;
        pa.n_in = n_in; pa.n_out = n;
        hipLaunchKernelGGL(fr_fft_pass_kernel, dim3(tiles), dim3(256), 0, ctx->stream, pa);
    }
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
// Decimation in time from BIT-REVERSED input (what bn254_fr_ntt_mont_dif leaves) to natural-order output, no gather either:
// out[k] = post[k] * sum_i in[bitrev(i)] w^(ik), k < n_out.  in may equal out.
int32_t bn254_fr_ntt_mont_from_bitrev(Ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n_out, uint32_t log_n, const uint64_t* tw, const uint64_t* post,
                                      const uint64_t scale_plain[4]) {
    const uint64_t n = 1ull << log_n;
    std::vector<uint32_t> ns;
    ns.push_back(std::min(10u, log_n));
    const uint32_t rem = log_n - ns[0], more = (rem + 6) / 7;
    for (uint32_t k = 0; k < more; k++) ns.push_back(rem / more + (k < rem % more ? 1 : 0));
    uint32_t s0 = 0;
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, n / 1024);
    // every pass reads and writes the same positions of its tile, so all of them but the last may run in place on the input ... which the
    // caller may want to keep: the first pass goes in -> out when n_out == n, otherwise the intermediate passes need a full-size buffer
    if (n_out != n && in != out) return ctx->fail(GL355_E_INVALID_ARG, "fr_ntt_from_bitrev: a truncated output needs the transform in place");
    for (size_t k = 0; k < ns.size(); k++) {
        FrPass pa;
        memset(&pa, 0, sizeof pa);
        pa.first = k == 0; pa.last = k + 1 == ns.size();
        pa.in = pa.first ? in : out;
        pa.out = out;
        pa.tw = tw; pa.log_n = log_n; pa.s0 = s0; pa.ns = ns[k];
        pa.in_mont = 1; pa.out_mont = 1; pa.no_gather = 1;
        if (scale_plain) { pa.use_scale = 1; pa.scale = h_to_mont(h_from_words(scale_plain)); }
        pa.n_in = n; pa.n_out = pa.last ? n_out : n;
        pa.post = pa.last ? post : nullptr;
        hipLaunchKernelGGL(fr_fft_pass_kernel, dim3(tiles), dim3(256), 0, ctx->stream, pa);
        s0 += ns[k];
    }
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
int32_t bn254_fr_ntt_mont(Ctx* ctx, const uint64_t* in, uint64_t n_in, uint64_t* out, uint64_t n_out, uint32_t log_n, const uint64_t* tw,
                          const uint64_t* pre, const uint64_t* post, const uint64_t scale_plain[4] /* or null */, uint64_t* work) {
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) {
        GL355_HIP(ctx, hipMemcpyAsync(out, in, 32, hipMemcpyDeviceToDevice, ctx->stream));        // (pre / post of a 1-point transform: not needed by any caller)
        return GL355_OK;
    }
    std::vector<uint32_t> ns;
    ns.push_back(std::min(10u, log_n));
    const uint32_t rem = log_n - ns[0], more = (rem + 6) / 7;
    for (uint32_t k = 0; k < more; k++) ns.push_back(rem / more + (k < rem % more ? 1 : 0));
    uint32_t s0 = 0;
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, n / 1024);
    for (size_t k = 0; k < ns.size(); k++) {
        FrPass pa;
        memset(&pa, 0, sizeof pa);
        pa.first = k == 0; pa.last = k + 1 == ns.size();
        pa.in = pa.first ? in : work;
        pa.out = (pa.last && !(pa.first && in == out)) ? out : work;
        pa.tw = tw; pa.log_n = log_n; pa.s0 = s0; pa.ns = ns[k];
        pa.in_mont = 1; pa.out_mont = 1;
        if (scale_plain) { pa.use_scale = 1; pa.scale = h_to_mont(h_from_words(scale_plain)); }
        pa.n_in = n_in; pa.n_out = pa.out == out ? n_out : n;
        pa.pre = pre;
        pa.post = pa.out == out ? post : nullptr;
        if (pa.out != out) pa.use_scale = 0;
        hipLaunchKernelGGL(fr_fft_pass_kernel, dim3(tiles), dim3(256), 0, ctx->stream, pa);
        s0 += ns[k];
    }
    if (ns.size() == 1 && in == out) {
        // single pass in place: the pass wrote `work` without post / scale (bit-reversed reads cannot run in place): apply them in the copy
        hipLaunchKernelGGL(fr_scale_copy_kernel, dim3((uint32_t)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t*)work, out, n_out, post,
                           scale_plain ? h_to_mont(h_from_words(scale_plain)) : to_u256(H256{{0, 0, 0, 0}}), scale_plain ? 1 : 0);
    }
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}
}  // namespace gl355

extern "C" {

// in: n_in = 2^log_in values, out: n_out values of the 2^log_n-point transform; shift == nullptr: the plain transform
static int32_t fr_ntt_run(Ctx* ctx, const uint64_t* in, uint32_t log_in, uint64_t* out, uint64_t n_out, uint32_t log_n, int32_t inverse,
                          const uint64_t* shift) {
    const uint64_t n = 1ull << log_n, n_in = 1ull << log_in;
    // omega_n = ROOT^(2^(28 - log_n)) (or its inverse), as a plain integer, then to Montgomery form: * R mod r
    H256 w = inverse ? H256{{BN254C_FR_ROOT_INV_64[0], BN254C_FR_ROOT_INV_64[1], BN254C_FR_ROOT_INV_64[2], BN254C_FR_ROOT_INV_64[3]}}
                     : H256{{BN254C_FR_ROOT_64[0], BN254C_FR_ROOT_64[1], BN254C_FR_ROOT_64[2], BN254C_FR_ROOT_64[3]}};
    for (uint32_t k = log_n; k < BN254C_FR_S; k++) w = h_mulmod(w, w);
    const H256 Rm = {{BN254C_FR_ONE_64[0], BN254C_FR_ONE_64[1], BN254C_FR_ONE_64[2], BN254C_FR_ONE_64[3]}};      // R mod r
    const u256 w_mont = to_u256(h_mulmod(w, Rm));
    const H256 e_inv = {{HR[0] - 2, HR[1], HR[2], HR[3]}};
    H256 scale_h = {{1, 0, 0, 0}};
    if (inverse) scale_h = h_powmod(H256{{n, 0, 0, 0}}, e_inv);              // n^-1 mod r, plain
    const u256 scale = to_u256(scale_h);
    // coset: powers of the shift multiply the inputs of the forward form, powers of its inverse (and 1/n) the outputs of the inverse form
    u256 pow_base = to_u256(H256{{0, 0, 0, 0}});
    if (shift) {
        H256 sh = {{shift[0], shift[1], shift[2], shift[3]}};
        while (h_geq(sh)) { unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 dd = (unsigned __int128)sh.l[i] - HR[i] - (uint64_t)br; sh.l[i] = (uint64_t)dd; br = (dd >> 64) & 1; } }
        if ((sh.l[0] | sh.l[1] | sh.l[2] | sh.l[3]) == 0) return ctx->fail(GL355_E_INVALID_ARG, "bn254_fr_coset_ntt: the shift must not be zero");
        if (inverse) sh = h_powmod(sh, e_inv);
        pow_base = to_u256(h_mulmod(sh, Rm));
    }
    const uint64_t n_pow = shift ? (inverse ? n_out : n_in) : 0;
    Scratch tw(ctx);
    const uint64_t n_hi = (std::max(n / 2, n_pow) + 1023) / 1024;
    GL355_TRY(tw.get((n / 2) * 32 + 32 + n * 32 + 2 * (1024 + n_hi) * 32 + n_pow * 32));
    uint64_t* twp = tw.as<uint64_t>();
    uint64_t* work = twp + 4 * (n / 2) + 4;                 // the first pass reads the data bit-reversed: it cannot run in place
    uint64_t* tw_lo = work + 4 * n;
    uint64_t* tw_hi = tw_lo + 4 * 1024;
    uint64_t* pw_lo = tw_hi + 4 * n_hi;
    uint64_t* pw_hi = pw_lo + 4 * 1024;
    uint64_t* pw = pw_hi + 4 * n_hi;
    const uint32_t hblk = (uint32_t)((n / 2 + 255) / 256);
    {
        ProfScope ps(ctx, "bn254_fr_ntt", (n_in + n_out) * 32);
        hipLaunchKernelGGL(fr_twiddle_seed_kernel, dim3((uint32_t)((1024 + n_hi + 255) / 256)), dim3(256), 0, ctx->stream, tw_lo, tw_hi, n_hi, w_mont);
        hipLaunchKernelGGL(fr_twiddle_kernel, dim3(hblk ? hblk : 1), dim3(256), 0, ctx->stream, twp, n / 2, tw_lo, tw_hi);
        if (n_pow) {
            hipLaunchKernelGGL(fr_twiddle_seed_kernel, dim3((uint32_t)((1024 + n_hi + 255) / 256)), dim3(256), 0, ctx->stream, pw_lo, pw_hi, n_hi, pow_base);
            if (inverse) hipLaunchKernelGGL(fr_power_plain_kernel, dim3((uint32_t)((n_pow + 255) / 256)), dim3(256), 0, ctx->stream, pw, n_pow, pw_lo, pw_hi, scale);
            else hipLaunchKernelGGL(fr_twiddle_kernel, dim3((uint32_t)((n_pow + 255) / 256)), dim3(256), 0, ctx->stream, pw, n_pow, pw_lo, pw_hi);
        }
        {
            // stages per pass: ten in the first (contiguous blocks), the rest in passes of at most six
            std::vector<uint32_t> ns;
            ns.push_back(std::min(10u, log_n));
            const uint32_t rem = log_n - ns[0], more = (rem + 6) / 7;
            for (uint32_t k = 0; k < more; k++) ns.push_back(rem / more + (k < rem % more ? 1 : 0));
            uint32_t s0 = 0;
            const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, n / 1024);
            for (size_t k = 0; k < ns.size(); k++) {
                FrPass pa;
                memset(&pa, 0, sizeof pa);
                pa.first = k == 0; pa.last = k + 1 == ns.size();
                pa.in = pa.first ? in : work;
                // the last pass may write straight to `out` unless it is also the first one and out aliases in (bit-reversed reads)
                pa.out = (pa.last && !(pa.first && in == out)) ? out : work;
                pa.tw = twp; pa.log_n = log_n; pa.s0 = s0; pa.ns = ns[k];
                pa.use_scale = inverse ? 1 : 0; pa.scale = scale;
                pa.n_in = n_in; pa.n_out = pa.out == out ? n_out : n;
                pa.pre = (shift && !inverse) ? pw : nullptr;
                pa.post = (shift && inverse && pa.out == out) ? pw : nullptr;
                hipLaunchKernelGGL(fr_fft_pass_kernel, dim3(tiles), dim3(256), 0, ctx->stream, pa);
                s0 += ns[k];
            }
            if (ns.size() == 1 && in == out) {
                if (shift && inverse) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_fr_coset_ntt: in-place inverse coset transform of <= 1024 points");
                GL355_HIP(ctx, hipMemcpyAsync(out, work, n_out * 32, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        GL355_HIP(ctx, hipGetLastError());
    }
    return GL355_OK;
}

int32_t gl355_bn254_fr_ntt(gl355_ctx* h, uint64_t* data, uint32_t log_n, int32_t inverse) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!data) return ctx->fail(GL355_E_INVALID_ARG, "bn254_fr_ntt: null data");
    if (log_n > 26) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_fr_ntt: log_n > 26 unsupported (Fr has 2-adicity 28)");
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) return GL355_OK;
    Staged sd(ctx);
    GL355_TRY(sd.open(data, n * 32, 3));
    GL355_TRY(fr_ntt_run(ctx, sd.as<uint64_t>(), log_n, sd.as<uint64_t>(), n, log_n, inverse, nullptr));
    return sd.finish();
}

int32_t gl355_bn254_fr_coset_ntt(gl355_ctx* h, const uint64_t* in, uint32_t log_small, uint32_t log_n, const uint64_t shift[4], int32_t inverse,
                                 uint64_t* out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!in || !out || !shift) return ctx->fail(GL355_E_INVALID_ARG, "bn254_fr_coset_ntt: null argument");
    if (in == out) return ctx->fail(GL355_E_INVALID_ARG, "bn254_fr_coset_ntt: in and out must differ");
    if (log_n > 26 || log_small > log_n) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_fr_coset_ntt: needs log_small <= log_n <= 26");
    if (log_n == 0) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_fr_coset_ntt: log_n == 0");
    const uint64_t n = 1ull << log_n, ns = 1ull << log_small;
    // forward: 2^log_small coefficients -> 2^log_n evaluations on shift * <omega_n>; inverse: 2^log_n evaluations -> 2^log_small coefficients
    const uint64_t n_in = inverse ? n : ns, n_out = inverse ? ns : n;
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(in, n_in * 32, 1));
    GL355_TRY(so.open(out, n_out * 32, 2));
    GL355_TRY(fr_ntt_run(ctx, si.as<uint64_t>(), inverse ? log_n : log_small, so.as<uint64_t>(), n_out, log_n, inverse, shift));
    return so.finish();
}

}  // extern "C"
namespace gl355 {
// max_bits: every scalar of the call is below 2^max_bits (256: no promise).  Windows above that hold only zero digits: they are not built,
// sorted or reduced (range-check columns are 16-bit values, the arithmetic chip's operands 64-bit: 2 and 5 windows of 20 bits instead of 13)
}
extern "C" {
static int32_t msm_run(gl355_ctx* h, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t m, uint64_t* result) {
    return bn254_msm_bits(h, points, scalars, n, m, 256, result);
}
}  // extern "C"
int32_t gl355::bn254_msm_bits(gl355_ctx* h, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t m, uint32_t max_bits, uint64_t* result,
                              const gl355_msm_bases* bases) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (bases && (bases->ctx != ctx || bases->n != n)) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm: prepared bases of another context or size");
    if (!result || ((!(points || bases) || !scalars) && n)) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm: null argument");
    if (n > (1ull << 26)) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_g1_msm: more than 2^26 points");
    if (m == 0) return GL355_OK;
    if (m > 64 || (uint64_t)m * n > (1ull << 27)) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_g1_msm_batch: more than 64 scalar sets or 2^27 scalars in all");
    const bool dev_result = ptr_is_device(result);
    if (n == 0) {
        if (dev_result) { GL355_HIP(ctx, hipMemsetAsync(result, 0, 64ull * m, ctx->stream)); GL355_HIP(ctx, ctx->wait()); }
        else memset(result, 0, 64ull * m);
        return GL355_OK;
    }
    uint32_t lg = 0;
    while ((1ull << lg) < n) lg++;
    MsmArgs a;
    memset(&a, 0, sizeof a);
    a.n = n;
    // window bits.  Fewer, larger windows mean fewer additions in the bucket phase (n per window) and more buckets to reduce; and the
    // TOP window should not be nearly empty: scalars are < r < 2^254, so a top window of only a few bits puts everything into a handful
    // of buckets (workgroup path below, contended counters).  254 = 14 * 17 + 16 = 12 * 20 + 14.  Measured (uniform scalars, ms):
    //   2^18: c = 15 / 16 / 17 -> 3.7 / 3.9 / 4.1;   2^20: 16 / 17 / 18 -> 7.3 / 7.1 / 8.3;   2^22: 16 / 17 / 18 / 19 -> 23.6 / 19.1 / 20.7 / 32.9
    //   2^23-point calls, k = 23 proof: c = 18 / 19 / 20 / 21 -> 1.155 / 1.115 / 1.118 / 1.235 s
    a.c = bases ? bases->c : (lg <= 6 ? 4 : (lg <= 18 ? lg - 2 : (lg <= 22 ? 17 : 20)));
    // scalars shorter than a window (range-check limbs): ONE window just wide enough that no digit reaches 2^(c-1), so nothing is negative, nothing carries
    // and the carry window does not exist -- 2^16 buckets for a 16-bit column instead of two windows of 2^19 (two-level sort from 12 bits on)
    const bool one_window = !bases && max_bits + 1 < a.c && max_bits + 1 >= 12;
    if (one_window) a.c = max_bits + 1;
    a.cb = a.c - 1;
    a.wps = 256 / a.c + 1;                                       // signed digits: the carry out of bit 255 needs a window of its own
    if (max_bits < 256) a.wps = std::min(a.wps, (std::max(1u, max_bits) + a.c - 1) / a.c + 1);
    if (one_window) a.wps = 1;
    if (bases && a.wps > bases->wps) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm: prepared bases hold too few windows");
    a.n_sets = m;
    a.n_windows = a.wps * m;
    // SHARED BUCKETS (prepared bases): point i of window w is the table entry w n + i, whose own digit is the scalar's w-th -- an MSM of wps n
    // points with ONE window per scalar set.  The digit array [set][w][i] is already that MSM's [set][w n + i], so past msm_digits_kernel every
    // kernel runs unchanged on the virtual sizes: 2^cb buckets per SET to size-sort, accumulate and reduce instead of per window, and the set's
    // sum comes out of the last level (no doublings between windows on the host).
    const uint32_t real_wps = a.wps;
    const uint64_t real_n = n;
    if (bases) {
        if (n * real_wps >= (1ull << 31)) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_g1_msm: prepared bases x windows beyond 2^31 entries");
        a.have_table = 1;
    }
    const uint64_t nb = 1ull << a.cb;
    uint64_t W = a.n_windows;
    Staged sp(ctx), ss(ctx);
    if (!bases) GL355_TRY(sp.open(points, n * 64, 1));
    GL355_TRY(ss.open(scalars, (uint64_t)m * n * 32, 1));
    a.points = bases ? nullptr : sp.as<uint64_t>(); a.scalars = ss.as<uint64_t>();
    // from here on the sizes are the virtual ones when the bases are prepared (one window of wps n points per set)
    const uint64_t rn = real_n;                                  // msm_digits_kernel alone runs on the real sizes
    if (bases) { n = real_n * real_wps; W = m; }
    // reduction levels: groups of 8 items, the last level takes what is left
    struct Lv { uint32_t t_in, kbits, shift; };
    std::vector<Lv> levels;
    uint64_t lvl_words = 0;
    for (uint32_t t = (uint32_t)nb, shift = 0; t > 1;) {
        uint32_t kb = 3;
        while ((1u << kb) > t) kb--;
        levels.push_back({t, kb, shift});
        t >>= kb; shift += kb;
        lvl_words += 2ull * W * t * 24;
    }
    // the two-level sort: from 2^11 buckets per window on (below that the histograms are small and the point count with them)
    const bool two_level = a.cb > MSM_FINE_BITS;      // (below: one-level sort with device-scope atomics per point and window)
    a.cbits = two_level ? a.cb - MSM_FINE_BITS : 0;
    const uint64_t nbin = 1ull << a.cbits;
    a.chunk = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, 32 * nbin), 1ull << 20);
    // WINDOW CHUNKS ON TWO STREAMS (measured and not used: K stays 1; the chunk machinery below is kept general).  The windows are independent until the host combines them, and an MSM's
    // phases are bound by different things: the sort by memory (scattered 4- and 8-byte writes), bucket accumulation by the VALU (0.81 of its
    // issue rate), the bucket reduction by latency -- at k = 23 the sort and the reduction are 11.5 % and 10.3 % of a proof's kernel time
    // next to 21.5 % of accumulation.  With K > 1 chunks the windows are cut into K chunks that alternate between the
    // context's stream and a second one, chunk i + 1 starting its sort when chunk i has finished its own, so that sort (i + 1) could run
    // under accumulate (i) and reduce (i) under accumulate (i + 1).  The k = 23 proof: K = 1 / 2 / 4 / 6 -> 1.143 / 1.142 / 1.151 / 1.207 s:
    // the accumulation kernel's waves hold the CUs' registers, the other stream's kernels get in only as it drains, and the smaller
    // launches lose what the overlap gains.  One chunk on one stream stays the default.
    const uint32_t K = 1;
    std::vector<uint32_t> w_lo(K + 1);
    for (uint32_t i = 0; i <= K; i++) w_lo[i] = (uint32_t)(W * i / K);
    // big-bucket work list: a bucket of sz > MSM_BIG points makes ceil(sz / slice) items with slice >= MSM_BIG_WG_POINTS, so over all
    // buckets at most W n / MSM_BIG_WG_POINTS + (number of big buckets) items, and at most W n / MSM_BIG big buckets (per chunk: its windows)
    std::vector<uint32_t> c_max_big(K), c_max_items(K);
    uint64_t tot_items = 0, tot_big = 0, lvl_words_all = 0;
    std::vector<uint64_t> c_lvl_words(K);
    for (uint32_t i = 0; i < K; i++) {
        const uint64_t Wc = w_lo[i + 1] - w_lo[i];
        c_max_big[i] = (uint32_t)(Wc * n / MSM_BIG + 1);
        c_max_items[i] = (uint32_t)(Wc * n / MSM_MID_SLICE + c_max_big[i] + 1);   // (mid-size buckets: items of MSM_MID_SLICE points)
        tot_items += c_max_items[i]; tot_big += c_max_big[i];
        c_lvl_words[i] = lvl_words / W * Wc;
        lvl_words_all += c_lvl_words[i];
    }
    const uint64_t fb_max_reg = W * n / MSM_FINE_BIG + 1, fb_max_items = W * n / MSM_FINE_SLICE + fb_max_reg + 1;      // (sums over all pairs)
    const uint64_t sort_words = two_level ? 3 * W * n + 3 * W * nbin + 2 + 2 + 2 * fb_max_reg + 2 * fb_max_items : 0;
    Scratch buf(ctx);
    const uint64_t words32 = (bases ? 0 : n * 16) + 3 * W * nb + W * n + W * nb * 24 + (uint64_t)K * (MSM_SIZE_BINS + 4) + lvl_words_all + 64 +
                             2ull * tot_items + 3ull * tot_big + 24ull * tot_items + sort_words;
    GL355_TRY(buf.get(words32 * 4 + 64));
    uint32_t* p = buf.as<uint32_t>();
    if (bases) a.pm = bases->tab; else { a.pm = p; p += n * 16; }
    a.hist = p; p += W * nb;
    uint32_t* small = p; p += (uint64_t)K * (MSM_SIZE_BINS + 4);  // per chunk: size histogram + the two work-list counters, cleared with the histograms
    a.cursor = p; p += W * nb;
    a.order = p; p += W * nb;
    a.idx = p; p += W * n;
    a.buckets = p; p += W * nb * 24;
    uint32_t* items_all = p; p += 2ull * tot_items;
    uint32_t* bigb_all = p; p += 3ull * tot_big;
    uint32_t* partial_all = p; p += 24ull * tot_items;
    uint32_t* lvl = p; p += lvl_words_all;
    if (two_level) {
        p += (2 - ((uintptr_t)p / 4) % 2) % 2;                      // 8-byte alignment of the pairs
        a.pairs = p; p += 2 * W * n;
        a.dig = p; p += W * n;
        a.coarse_cnt = p; p += W * nbin;
        a.coarse_fill = p; p += W * nbin;
        a.coarse_start = p; p += W * nbin;
        a.fb_counters = p; p += 2;
        a.fb_regions = p; p += 2 * fb_max_reg;
        a.fb_items = p; p += 2 * fb_max_items;
        a.fb_max_reg = (uint32_t)fb_max_reg; a.fb_max_items = (uint32_t)fb_max_items;
        GL355_HIP(ctx, hipMemsetAsync(a.coarse_cnt, 0, 2 * W * nbin * 4, ctx->stream));
        GL355_HIP(ctx, hipMemsetAsync(a.fb_counters, 0, 8, ctx->stream));
    }
    GL355_HIP(ctx, hipMemsetAsync(a.hist, 0, (W * nb + (uint64_t)K * (MSM_SIZE_BINS + 4)) * 4, ctx->stream));
    const uint32_t blk = (uint32_t)((rn + 255) / 256);
    std::vector<const uint32_t*> fin_s(K), fin_w(K, nullptr);
    std::vector<const uint32_t*> counters(K);
    {
        ProfScope ps(ctx, "bn254_g1_msm", n * (64 + 32ull * m));
        hipStream_t st[2] = {ctx->stream, ctx->stream};
        hipEvent_t ev_ready = nullptr;
        if (two_level) hipLaunchKernelGGL(msm_digits_kernel, dim3(blk), dim3(256), 0, ctx->stream, a);
        if (bases) { a.n = n; a.n_windows = (uint32_t)W; a.wps = 1; }      // the virtual MSM: every kernel below sees one window per set
        if (K > 1) {
            GL355_TRY(ctx->aux_stream_get(&st[1]));
            GL355_TRY(ctx->order_event(0, &ev_ready));
            GL355_HIP(ctx, hipEventRecord(ev_ready, ctx->stream));          // digits, cleared histograms: what every chunk starts from
            GL355_HIP(ctx, hipStreamWaitEvent(st[1], ev_ready, 0));
        }
        uint64_t off_items = 0, off_big = 0;
        for (uint32_t ci = 0; ci < K; ci++) {
            hipStream_t s_ = st[ci & 1];
            const uint64_t w0 = w_lo[ci], Wc = w_lo[ci + 1] - w0;
            MsmArgs c = a;                                                  // the chunk's windows as a job of its own: every per-window array shifted
            c.n_windows = (uint32_t)Wc;
            c.hist += w0 * nb; c.cursor += w0 * nb; c.order += w0 * nb; c.idx += w0 * n; c.buckets += w0 * nb * 24;
            c.size_hist = small + (uint64_t)ci * (MSM_SIZE_BINS + 4); c.big_counters = c.size_hist + MSM_SIZE_BINS;
            c.max_items = c_max_items[ci]; c.max_big = c_max_big[ci];
            c.big_items = items_all + 2 * off_items; c.big_partial = partial_all + 24 * off_items; c.big_buckets = bigb_all + 3 * off_big;
            off_items += c.max_items; off_big += c.max_big;
            counters[ci] = c.big_counters;
            if (two_level) { c.pairs += 2 * w0 * n; c.dig += w0 * n; c.coarse_cnt += w0 * nbin; c.coarse_fill += w0 * nbin; c.coarse_start += w0 * nbin; }
            const uint32_t bblk = (uint32_t)((Wc * nb + 255) / 256);
            if (ci > 0 && K > 1) {                                          // this chunk's sort starts when the previous chunk's has finished
                hipEvent_t e;
                GL355_TRY(ctx->order_event(ci, &e));
                GL355_HIP(ctx, hipStreamWaitEvent(s_, e, 0));
            }
            if (two_level) {
                const dim3 cgrid((uint32_t)((n + c.chunk - 1) / c.chunk), (uint32_t)Wc);
                hipLaunchKernelGGL(msm_coarse_count_kernel, cgrid, dim3(256), nbin * 4, s_, c);
                hipLaunchKernelGGL(msm_coarse_scan_kernel, dim3((uint32_t)Wc), dim3(1024), 0, s_, c);
                hipLaunchKernelGGL(msm_coarse_scatter_kernel, cgrid, dim3(256), nbin * 8, s_, c);
                hipLaunchKernelGGL(msm_fine_sort_kernel, dim3((uint32_t)nbin, (uint32_t)Wc), dim3(256), 0, s_, c);
                // regions too large for one workgroup (usually none: the three kernels behind the list then find empty work lists)
                hipLaunchKernelGGL(msm_fine_big_list_kernel, dim3((uint32_t)((Wc * nbin + 255) / 256)), dim3(256), 0, s_, c);
                hipLaunchKernelGGL(msm_fine_big_count_kernel, dim3(2048), dim3(256), 0, s_, c);
                hipLaunchKernelGGL(msm_fine_big_scan_kernel, dim3(256), dim3(256), 0, s_, c);
                hipLaunchKernelGGL(msm_fine_big_scatter_kernel, dim3(2048), dim3(256), 0, s_, c);
            } else {
                hipLaunchKernelGGL(msm_prepare_kernel, dim3(blk), dim3(256), 0, s_, c);
                hipLaunchKernelGGL(msm_scan_kernel, dim3((uint32_t)Wc), dim3(1024), 0, s_, c);
                hipLaunchKernelGGL(msm_scatter_kernel, dim3(blk), dim3(256), 0, s_, c);
            }
            if (ci + 1 < K) {
                hipEvent_t e;
                GL355_TRY(ctx->order_event(ci + 1, &e));
                GL355_HIP(ctx, hipEventRecord(e, s_));
            }
            hipLaunchKernelGGL(msm_size_hist_kernel, dim3(bblk), dim3(256), 0, s_, c);
            hipLaunchKernelGGL(msm_size_scan_kernel, dim3(1), dim3(64), 0, s_, c);
            hipLaunchKernelGGL(msm_order_kernel, dim3(bblk), dim3(256), 0, s_, c);
            hipLaunchKernelGGL(msm_big_list_kernel, dim3(bblk), dim3(256), 0, s_, c);
            hipLaunchKernelGGL(msm_bucket_kernel, dim3(bblk), dim3(256), 0, s_, c);
            hipLaunchKernelGGL(msm_mid_partial_kernel, dim3(std::min<uint32_t>((c.max_items + 255) / 256, 2048)), dim3(256), 0, s_, c);
            hipLaunchKernelGGL(msm_big_partial_kernel, dim3(std::min<uint32_t>(c.max_items, 1536)), dim3(256), 0, s_, c);
            hipLaunchKernelGGL(msm_mid_final_kernel, dim3(std::min<uint32_t>((c.max_big + 63) / 64, 1024)), dim3(64), 0, s_, c);
            hipLaunchKernelGGL(msm_big_final_kernel, dim3(std::min<uint32_t>(c.max_big, 512)), dim3(256), 0, s_, c);
            const uint32_t *cs = c.buckets, *cw = nullptr;
            uint32_t* lv_p = lvl;
            lvl += c_lvl_words[ci];
            for (const Lv& lv : levels) {
                MsmLevel l;
                l.in_s = cs; l.in_w = cw; l.t_in = lv.t_in; l.kbits = lv.kbits; l.shift = lv.shift; l.n_windows = (uint32_t)Wc;
                const uint64_t groups = lv.t_in >> lv.kbits;
                l.out_s = lv_p; lv_p += Wc * groups * 24;
                l.out_w = lv_p; lv_p += Wc * groups * 24;
                // eight lanes per group where a level is latency-bound (few groups); the large first levels are throughput-bound and the scan
                // costs them twice the wave-level additions.  coop_max: the largest W x groups that runs the cooperative form
                constexpr uint64_t coop_max = 16384;
                if (lv.kbits == 3 && Wc * groups <= coop_max) hipLaunchKernelGGL(msm_level_coop_kernel, dim3((uint32_t)((Wc * groups + 7) / 8)), dim3(64), 0, s_, l);
                else hipLaunchKernelGGL(msm_level_kernel, dim3((uint32_t)((Wc * groups + 63) / 64)), dim3(64), 0, s_, l);
                cs = l.out_s; cw = l.out_w;
            }
            fin_s[ci] = cs; fin_w[ci] = cw;
        }
        if (K > 1) {                                                        // the context's stream continues when the second one is done
            hipEvent_t e;
            GL355_TRY(ctx->order_event(K, &e));
            GL355_HIP(ctx, hipEventRecord(e, st[1]));
            GL355_HIP(ctx, hipStreamWaitEvent(ctx->stream, e, 0));
        }
        GL355_HIP(ctx, hipGetLastError());
    }
    // per window S (and Wt when there was at least one level): 2 x W Jacobian points to the host, which combines the windows
    std::vector<uint32_t> hs(W * 24), hw(W * 24, 0), big_used(2ull * K, 0);
    for (uint32_t ci = 0; ci < K; ci++) {
        const uint64_t w0 = w_lo[ci], Wc = w_lo[ci + 1] - w0;
        GL355_HIP(ctx, ctx->d2h(big_used.data() + 2 * ci, counters[ci], 8));
        GL355_HIP(ctx, ctx->d2h(hs.data() + w0 * 24, fin_s[ci], Wc * 96));
        if (fin_w[ci]) GL355_HIP(ctx, ctx->d2h(hw.data() + w0 * 24, fin_w[ci], Wc * 96));
    }
    GL355_HIP(ctx, ctx->wait());
    for (uint32_t ci = 0; ci < K; ci++)
        if (big_used[2 * ci] > c_max_items[ci] || big_used[2 * ci + 1] > c_max_big[ci]) return ctx->fail(GL355_E_HIP, "bn254_g1_msm: big-bucket work list overflow (internal bound)");
    const bool have_w = !levels.empty();
    std::vector<uint64_t> res(8ull * m);
    for (uint32_t set = 0; set < m; set++)
        bn254_g1_horner_host(hs.data() + 24ull * set * a.wps, have_w ? hw.data() + 24ull * set * a.wps : nullptr, a.wps, a.c, res.data() + 8 * set);
    if (dev_result) { GL355_HIP(ctx, hipMemcpyAsync(result, res.data(), 64ull * m, hipMemcpyHostToDevice, ctx->stream)); GL355_HIP(ctx, ctx->wait()); }
    else memcpy(result, res.data(), 64ull * m);
    return GL355_OK;
}

extern "C" {
int32_t gl355_bn254_g1_msm(gl355_ctx* h, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint64_t result[8]) {
    return msm_run(h, points, scalars, n, 1, result);
}
int32_t gl355_bn254_g1_msm_batch(gl355_ctx* h, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t n_sets, uint64_t* results) {
    return msm_run(h, points, scalars, n, n_sets, results);
}


// Window width of a prepared base set.  With the buckets shared by all windows the reduction is paid once per scalar set, so wider windows than the
// per-window form's 17 - 20 bits pay: fewer windows = fewer additions in the bucket loops (n per window), until the buckets outnumber them.
static uint32_t msm_prepared_window_bits(uint32_t lg) {
    // ... and the top window should not be nearly empty (r < 2^254: a top window of two bits is four buckets of millions of points): the widest c from
    // lg - 1 down whose top window keeps at least c / 3 bits.  2^23: 22 (11 windows of 22 bits + 12 bits), 2^22, 2^21: 20, 2^20: 19, 2^18: 17
    // (k = 23 proof, ms of MSM kernels: c = 20 / 21 / 22 / 23 -> 409 / 440 / 408 / 480 before the mid-size bucket items grew; 417 without tables)
    for (uint32_t c = std::min(22u, std::max(13u, lg) - 1); c > 12; c--)
        if (254 - c * (253 / c) >= (c + 2) / 3) return c;
    return 12;
}
int32_t gl355_bn254_g1_msm_prepare(gl355_ctx* h, const uint64_t* points, uint64_t n, gl355_msm_bases** out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!points || !out || !n) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm_prepare: null argument or no points");
    uint32_t lg = 0;
    while ((1ull << lg) < n) lg++;
    const uint32_t c = msm_prepared_window_bits(lg), wps = 256 / c + 1;
    if (n > (1ull << 26) || n * wps >= (1ull << 31)) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_g1_msm_prepare: too many points");
    std::unique_ptr<gl355_msm_bases> b(new gl355_msm_bases{ctx, nullptr, n, c, wps});
    void* tab = nullptr;
    GL355_TRY(ctx->alloc((size_t)wps * n * 64, &tab));
    b->tab = static_cast<uint32_t*>(tab);
    Staged sp(ctx);
    int32_t rc = sp.open(points, n * 64, 1);
    Scratch tmp(ctx);
    MsmTabArgs a;
    memset(&a, 0, sizeof a);
    a.points = sp.as<uint64_t>(); a.tab = b->tab; a.n = n; a.c = c; a.wps = wps;
    a.chunk = std::min<uint64_t>(n, 1ull << 20);
    if (rc == GL355_OK) rc = tmp.get((size_t)(wps - 1) * a.chunk * 64 + 64);
    if (rc != GL355_OK) { ctx->release(tab); return rc; }
    a.zs = tmp.as<uint32_t>(); a.pre = a.zs + 8ull * (wps - 1) * a.chunk;
    {
        ProfScope ps(ctx, "bn254_g1_msm_prepare", n * 64ull * (1 + wps));
        for (a.i0 = 0; a.i0 < n; a.i0 += a.chunk)
            hipLaunchKernelGGL(msm_table_build_kernel, dim3((uint32_t)((a.chunk + 255) / 256)), dim3(256), 0, ctx->stream, a);
    }
    if (hipGetLastError() != hipSuccess || ctx->wait() != hipSuccess) { ctx->release(tab); return ctx->fail(GL355_E_HIP, "bn254_g1_msm_prepare: kernel failed"); }
    *out = b.release();
    return GL355_OK;
}
int32_t gl355_bn254_g1_msm_prepared(gl355_ctx* h, const gl355_msm_bases* bases, const uint64_t* scalars, uint32_t n_sets, uint64_t* results) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!bases) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm_prepared: null bases");
    return bn254_msm_bits(h, nullptr, scalars, bases->n, n_sets, 256, results, bases);
}
int32_t gl355_bn254_g1_msm_bases_free(gl355_ctx* h, gl355_msm_bases* bases) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!bases) return GL355_OK;
    if (bases->ctx != ctx) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm_bases_free: bases of another context");
    (void)ctx->wait();
    ctx->release(bases->tab);
    delete bases;
    return GL355_OK;
}
int32_t gl355_bn254_g1_fixed_base_mul(gl355_ctx* h, const uint64_t base[8], const uint64_t* scalars, uint64_t n, uint64_t* out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!base || ((!scalars || !out) && n)) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_fixed_base_mul: null argument");
    if (n > (1ull << 26)) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_g1_fixed_base_mul: more than 2^26 scalars");
    if (n == 0) return GL355_OK;
    Staged sb(ctx), ss(ctx), so(ctx);
    GL355_TRY(sb.open(base, 64, 1));
    GL355_TRY(ss.open(scalars, n * 32, 1));
    GL355_TRY(so.open(out, n * 64, 2));
    Scratch buf(ctx);
    GL355_TRY(buf.get((32 * 24 + 32 * 256 * 16) * 4 + 64));
    FbArgs a;
    a.base = sb.as<uint64_t>(); a.scalars = ss.as<uint64_t>(); a.n = n; a.out = so.as<uint64_t>();
    a.win = buf.as<uint32_t>(); a.table = a.win + 32 * 24;
    {
        ProfScope ps(ctx, "bn254_g1_fixed_base_mul", n * 96);
        hipLaunchKernelGGL(fb_windows_kernel, dim3(1), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL(fb_table_kernel, dim3(32), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(fb_mul_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
        GL355_HIP(ctx, hipGetLastError());
    }
    return so.finish();
}


// ---- KZG composites ------------------------------------------------------------------------------------------------------------
// ParamsKZG::setup(k, rng) with the secret handed in (verifier_api.rs:77): g[i] = [tau^i] G1, g_lagrange[i] = [L_i(tau)] G1, G1 = (1, 2).
// halo2 computes g_lagrange by an inverse FFT over the GROUP; here the Lagrange scalars are evaluated in Fr (one batched inversion per 16
// points) and go through the same fixed-base kernel as the powers.  g_lagrange may be NULL.  Outputs are affine points (n x 8 words).
int32_t gl355_kzg_setup(gl355_ctx* h, const uint64_t tau[4], uint32_t log_n, uint64_t* g, uint64_t* g_lagrange) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!tau || !g) return ctx->fail(GL355_E_INVALID_ARG, "kzg_setup: null argument");
    if (log_n > 26) return ctx->fail(GL355_E_UNSUPPORTED, "kzg_setup: log_n > 26");
    const uint64_t n = 1ull << log_n;
    uint64_t tau_h[4];                                        // tau may be device memory like every other operand
    if (ptr_is_device(tau)) { GL355_HIP(ctx, hipMemcpy(tau_h, tau, 32, hipMemcpyDeviceToHost)); } else memcpy(tau_h, tau, 32);
    const H256 t = h_from_words(tau_h);
    const u256 tau_mont = h_to_mont(t);
    {   // tau in the 2^log_n domain <=> tau^n = 1: refused whether or not the Lagrange bases are asked for (the header says so)
        H256 tn = t;
        for (uint32_t k = 0; k < log_n; k++) tn = h_mulmod(tn, tn);
        if (tn.l[0] == 1 && (tn.l[1] | tn.l[2] | tn.l[3]) == 0) return ctx->fail(GL355_E_INVALID_ARG, "kzg_setup: tau lies in the evaluation domain");
    }
    Scratch sc(ctx);
    GL355_TRY(sc.get((g_lagrange ? 2 : 1) * n * 32 + 64));
    uint64_t* d_s = sc.as<uint64_t>();
    uint64_t* d_pre = d_s + 4 * n;                          // Lagrange pass only
    uint32_t* d_bad = reinterpret_cast<uint32_t*>(d_s + 4 * n * (g_lagrange ? 2 : 1));
    const uint64_t gen[8] = {1, 0, 0, 0, 2, 0, 0, 0};
    {
        ProfScope ps(ctx, "kzg_setup_scalars", n * 32);
        hipLaunchKernelGGL(kzg_tau_powers_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, tau_mont, n, d_s);
        GL355_HIP(ctx, hipGetLastError());
    }
    GL355_TRY(gl355_bn254_g1_fixed_base_mul(h, gen, d_s, n, g));
    if (g_lagrange) {
        // c = (tau^n - 1) / n
        H256 tn = t;
        for (uint32_t k = 0; k < log_n; k++) tn = h_mulmod(tn, tn);
        const H256 e_inv = {{HR[0] - 2, HR[1], HR[2], HR[3]}};
        const H256 c = h_mulmod(h_submod(tn, H256{{1, 0, 0, 0}}), h_powmod(H256{{n, 0, 0, 0}}, e_inv));
        GL355_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
        {
            ProfScope ps(ctx, "kzg_setup_scalars", n * 32);
            const uint64_t lanes = (n + KZG_LG_CHUNK - 1) / KZG_LG_CHUNK;
            const H256 w = h_root_of_unity(log_n);
            hipLaunchKernelGGL(kzg_lagrange_kernel, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, ctx->stream, tau_mont, h_to_mont(w),
                               h_to_mont(h_powmod(w, e_inv)), h_to_mont(c), n, d_s, d_pre, d_bad);
            GL355_HIP(ctx, hipGetLastError());
        }
        uint32_t bad = 0;
        GL355_HIP(ctx, ctx->d2h(&bad, d_bad, 4));
        GL355_HIP(ctx, ctx->wait());
        if (bad) return ctx->fail(GL355_E_INVALID_ARG, "kzg_setup: tau lies in the evaluation domain");
        GL355_TRY(gl355_bn254_g1_fixed_base_mul(h, gen, d_s, n, g_lagrange));
    }
    return GL355_OK;
}

// ParamsKZG::commit / commit_lagrange: result = sum_i poly[i] * g[i] over 2^log_n bases.  values_form = 0: `poly` are the scalars that go
// with the given bases as they are (coefficients with the monomial bases g, or evaluations with g_lagrange -- an MSM does not care);
// values_form = 1: `poly` are EVALUATIONS over the 2^log_n domain but `g` are the monomial bases: inverse FFT on a scratch copy, then the MSM.
int32_t gl355_kzg_commit(gl355_ctx* h, const uint64_t* g, const uint64_t* poly, uint32_t log_n, int32_t values_form, uint64_t result[8]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!g || !poly || !result) return ctx->fail(GL355_E_INVALID_ARG, "kzg_commit: null argument");
    if (log_n > 26) return ctx->fail(GL355_E_UNSUPPORTED, "kzg_commit: log_n > 26");
    const uint64_t n = 1ull << log_n;
    if (!values_form) return gl355_bn254_g1_msm(h, g, poly, n, result);
    Scratch sc(ctx);
    GL355_TRY(sc.get(n * 32));
    uint64_t* d_c = sc.as<uint64_t>();
    GL355_HIP(ctx, hipMemcpyAsync(d_c, poly, n * 32, ptr_is_device(poly) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (log_n) GL355_TRY(fr_ntt_run(ctx, d_c, log_n, d_c, n, log_n, 1, nullptr));
    return gl355_bn254_g1_msm(h, g, d_c, n, result);
}

}  // extern "C"

namespace gl355 {
// Q and E of the comment above the division kernels, levels chained on the stream; A and Q are device arrays (Q plain at the top level)
int32_t kzg_divide(Ctx* ctx, const uint64_t* A, uint64_t m, const H256& z, int a_is_mont, uint64_t* Q, int q_plain, uint64_t* E_mont) {
    const u256 z_mont = h_to_mont(z);
    if (m <= KZG_DIV_CHUNK) {
        hipLaunchKernelGGL(kzg_div_walk_kernel, dim3(1), dim3(64), 0, ctx->stream, A, m, z_mont, a_is_mont, (const uint64_t*)nullptr, Q, q_plain, E_mont);
        GL355_HIP(ctx, hipGetLastError());
        return GL355_OK;
    }
    const uint64_t chunks = (m + KZG_DIV_CHUNK - 1) / KZG_DIV_CHUNK;
    Scratch sc(ctx);
    GL355_TRY(sc.get(chunks * 64));
    uint64_t* H = sc.as<uint64_t>();
    uint64_t* C = H + 4 * chunks;
    hipLaunchKernelGGL(kzg_div_chunk_kernel, dim3((uint32_t)((chunks + 63) / 64)), dim3(64), 0, ctx->stream, A, m, z_mont, a_is_mont, H);
    GL355_HIP(ctx, hipGetLastError());
    H256 zc = z;                                             // Z = z^chunk
    for (uint32_t k = 1; k < KZG_DIV_CHUNK; k <<= 1) zc = h_mulmod(zc, zc);
    GL355_TRY(kzg_divide(ctx, H, chunks, zc, 1, C, 0, nullptr));
    hipLaunchKernelGGL(kzg_div_walk_kernel, dim3((uint32_t)((chunks + 63) / 64)), dim3(64), 0, ctx->stream, A, m, z_mont, a_is_mont, (const uint64_t*)C, Q, q_plain, E_mont);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

}  // namespace gl355

extern "C" {

// The single-point KZG opening (what halo2's multiopen provers reduce to per rotation set): eval = p(z), witness = commit((p - p(z)) / (X - z)).
// `coeffs` are the 2^log_n coefficients of p, `g` the monomial bases.  quotient (optional, 2^log_n x 4 words, device or host) receives the
// quotient's coefficients (the last one is 0).
int32_t gl355_kzg_open(gl355_ctx* h, const uint64_t* g, const uint64_t* coeffs, uint32_t log_n, const uint64_t z[4], uint64_t eval[4],
                       uint64_t witness[8], uint64_t* quotient) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!g || !coeffs || !z || !eval || !witness) return ctx->fail(GL355_E_INVALID_ARG, "kzg_open: null argument");
    if (log_n > 26) return ctx->fail(GL355_E_UNSUPPORTED, "kzg_open: log_n > 26");
    const uint64_t n = 1ull << log_n;
    Staged sp(ctx);
    GL355_TRY(sp.open(coeffs, n * 32, 1));
    Scratch sq(ctx);
    GL355_TRY(sq.get(n * 32 + 64));
    uint64_t* d_q = sq.as<uint64_t>();
    uint64_t* d_e = d_q + 4 * n;
    {
        ProfScope ps(ctx, "kzg_divide", n * 64);
        uint64_t z_h[4];
        if (ptr_is_device(z)) { GL355_HIP(ctx, hipMemcpy(z_h, z, 32, hipMemcpyDeviceToHost)); } else memcpy(z_h, z, 32);
        GL355_TRY(kzg_divide(ctx, sp.as<uint64_t>(), n, h_from_words(z_h), 0, d_q, 1, d_e));
        hipLaunchKernelGGL(kzg_from_mont1_kernel, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t*)d_e, d_e + 4);
        GL355_HIP(ctx, hipGetLastError());
    }
    if (ptr_is_device(eval)) GL355_HIP(ctx, hipMemcpyAsync(eval, d_e + 4, 32, hipMemcpyDeviceToDevice, ctx->stream));
    else GL355_HIP(ctx, ctx->d2h(eval, d_e + 4, 32));
    if (quotient) GL355_HIP(ctx, hipMemcpyAsync(quotient, d_q, n * 32, ptr_is_device(quotient) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    GL355_HIP(ctx, ctx->wait());
    return gl355_bn254_g1_msm(h, g, d_q, n, witness);
}

}  // extern "C"
