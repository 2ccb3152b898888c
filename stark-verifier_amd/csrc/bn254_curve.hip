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
#include "gl355_internal.h"

#define BN254C_QUAL __device__ __constant__ const
#include "bn254_curve_tables.h"

namespace gl355 {

struct u256 { uint32_t l[8]; };
enum { F_R = 0, F_Q = 1 };

template <int F> GL_DEV const uint32_t* f_mod() { return F == F_Q ? BN254C_FQ_MOD : BN254C_FR_MOD; }
template <int F> GL_DEV const uint32_t* f_two_mod() { return F == F_Q ? BN254C_FQ_TWO_MOD : BN254C_FR_TWO_MOD; }
template <int F> GL_DEV const uint32_t* f_r2() { return F == F_Q ? BN254C_FQ_R2 : BN254C_FR_R2; }
template <int F> GL_DEV const uint32_t* f_one() { return F == F_Q ? BN254C_FQ_ONE : BN254C_FR_ONE; }

GL_DEV u256 u_const(const uint32_t* p) {
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = p[j];
    return r;
}
GL_DEV u256 u_zero() {
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = 0;
    return r;
}
GL_DEV bool u_is_zero(const u256& a) {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) o |= a.l[j];
    return o == 0;
}
GL_DEV bool u_eq(const u256& a, const u256& b) {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) o |= a.l[j] ^ b.l[j];
    return o == 0;
}
// a - m if a >= m else a
GL_DEV u256 u_cond_sub(const u256& a, const uint32_t* m) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t v = (uint64_t)a.l[j] - m[j] - br;
        d[j] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = br ? a.l[j] : d[j];
    return r;
}
// a * b * R^-1 (mod m), result < 2m for a, b < 2m
template <int F>
__device__ __noinline__ u256 m_mul(u256 a, u256 b) {
    const uint32_t* M = f_mod<F>();
    const uint32_t n0 = F == F_Q ? BN254C_FQ_N0INV : BN254C_FR_N0INV;
    uint32_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t t9 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = (uint64_t)a.l[j] * b.l[i] + ((uint64_t)t[j] + c);
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t9 = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * n0;
        c = ((uint64_t)m * M[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c = (uint64_t)m * M[j] + ((uint64_t)t[j] + c);
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t9 + (uint32_t)(c >> 32);
    }
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = t[j];
    return r;
}
template <int F> GL_DEV u256 m_add(const u256& a, const u256& b) {
    u256 s;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.l[j] + b.l[j];
        s.l[j] = (uint32_t)c;
        c >>= 32;
    }
    return u_cond_sub(s, f_two_mod<F>());
}
template <int F> GL_DEV u256 m_sub(const u256& a, const u256& b) {       // a + (2m - b), both < 2m
    const uint32_t* tm = f_two_mod<F>();
    u256 nb;
    uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t v = (uint64_t)tm[j] - b.l[j] - br;
        nb.l[j] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    return m_add<F>(a, nb);
}
template <int F> GL_DEV u256 m_canon(const u256& a) { return u_cond_sub(a, f_mod<F>()); }       // < 2m -> < m
template <int F> GL_DEV bool m_is_zero(const u256& a) { return u_is_zero(m_canon<F>(a)); }
template <int F> GL_DEV bool m_eq(const u256& a, const u256& b) { return u_eq(m_canon<F>(a), m_canon<F>(b)); }
// any 256-bit integer -> Montgomery form (< 2m): 2^256 < 6m, so five conditional subtractions bring the input below m first
template <int F> GL_DEV u256 m_from_int(u256 a) {
#pragma unroll 1
    for (int k = 0; k < 5; k++) a = u_cond_sub(a, f_mod<F>());
    return m_mul<F>(a, u_const(f_r2<F>()));
}
template <int F> GL_DEV u256 m_to_int(const u256& a) {
    u256 one = u_zero();
    one.l[0] = 1;
    return m_canon<F>(m_mul<F>(a, one));
}
GL_DEV u256 load256(const uint64_t* p) {
    u256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)p[i]; r.l[2 * i + 1] = (uint32_t)(p[i] >> 32); }
    return r;
}
GL_DEV void store256(uint64_t* p, const u256& a) {
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
}
template <int F> GL_DEV u256 m_pow_u64(u256 a, uint64_t e) {
    u256 r = u_const(f_one<F>());
    while (e) {
        if (e & 1) r = m_mul<F>(r, a);
        a = m_mul<F>(a, a);
        e >>= 1;
    }
    return r;
}
// a^(m-2): the inverse
template <int F> GL_DEV u256 m_inv(const u256& a) {
    const uint32_t* M = f_mod<F>();
    u256 r = u_const(f_one<F>());
#pragma unroll 1
    for (int i = 255; i >= 0; i--) {
        r = m_mul<F>(r, r);
        uint32_t w = M[i >> 5];
        if ((i >> 5) == 0) w -= 2;               // low limb of both primes is > 2: no borrow
        if ((w >> (i & 31)) & 1) r = m_mul<F>(r, a);
    }
    return r;
}

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
// tw[i] = w^i (Montgomery), i < count
__global__ void fr_twiddle_kernel(uint64_t* tw, uint64_t count, u256 w_mont) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    store256(tw + 4 * i, m_pow_u64<F_R>(w_mont, i));
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
    uint32_t c, n_windows;      // window bits, windows
    uint32_t* pm;               // [n][16] points in Montgomery form
    uint32_t* hist;             // [W][2^c]      counts, then exclusive offsets
    uint32_t* cursor;           // [W][2^c]      scatter cursors
    uint32_t* idx;              // [W][n]        point indices sorted by digit
    uint32_t* buckets;          // [W][2^c][24]  Jacobian bucket sums
    uint32_t* partial;          // [W][chunks][24]
    uint32_t* wsum;             // [W][24]
    uint32_t chunk;             // buckets per lane in the window reduction
    uint64_t* result;           // [8]
};
GL_DEV uint32_t msm_digit(const uint64_t* k, uint32_t w, uint32_t c) {
    const uint32_t bit = w * c;
    if (bit >= 256) return 0;
    const uint32_t limb = bit >> 6, off = bit & 63;
    uint64_t v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (uint32_t)(v & ((1ull << c) - 1));
}
__global__ void msm_prepare_kernel(MsmArgs a) {          // points to Montgomery form + digit histograms
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const u256 x = load256(a.points + 8 * i), y = load256(a.points + 8 * i + 4);
    const bool ident = u_is_zero(x) && u_is_zero(y);
    const u256 xm = ident ? u_zero() : m_from_int<F_Q>(x), ym = ident ? u_zero() : m_from_int<F_Q>(y);
    uint32_t* d = a.pm + 16 * i;
#pragma unroll
    for (int j = 0; j < 8; j++) { d[j] = xm.l[j]; d[8 + j] = ym.l[j]; }
    if (ident) return;
    const uint64_t* k = a.scalars + 4 * i;
    for (uint32_t w = 0; w < a.n_windows; w++) {
        const uint32_t dg = msm_digit(k, w, a.c);
        if (dg) atomicAdd(a.hist + ((uint64_t)w << a.c) + dg, 1u);
    }
}
// per window: exclusive scan of the 2^c counts (one workgroup), offsets copied to the cursors
__global__ void __launch_bounds__(1024) msm_scan_kernel(MsmArgs a) {
    __shared__ uint32_t sh[1024];
    const uint32_t w = blockIdx.x, nb = 1u << a.c, tid = threadIdx.x;
    uint32_t* h = a.hist + ((uint64_t)w << a.c);
    uint32_t* cur = a.cursor + ((uint64_t)w << a.c);
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
    if (i >= a.n) return;
    const uint32_t* d = a.pm + 16 * i;
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) o |= d[j];
    if (!o) return;                                       // the identity contributes nothing
    const uint64_t* k = a.scalars + 4 * i;
    for (uint32_t w = 0; w < a.n_windows; w++) {
        const uint32_t dg = msm_digit(k, w, a.c);
        if (!dg) continue;
        const uint32_t pos = atomicAdd(a.cursor + ((uint64_t)w << a.c) + dg, 1u);
        a.idx[(uint64_t)w * a.n + pos] = (uint32_t)i;
    }
}
// one lane per (window, bucket): sum of the bucket's points
__global__ void __launch_bounds__(128) msm_bucket_kernel(MsmArgs a) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (b >= (1u << a.c)) return;
    jac acc = j_identity();
    if (b) {
        const uint32_t lo = a.hist[((uint64_t)w << a.c) + b], hi = a.cursor[((uint64_t)w << a.c) + b];   // cursor = end after the scatter
        for (uint32_t k = lo; k < hi; k++) {
            const uint32_t* p = a.pm + 16ull * a.idx[(uint64_t)w * a.n + k];
            u256 x, y;
#pragma unroll
            for (int j = 0; j < 8; j++) { x.l[j] = p[j]; y.l[j] = p[8 + j]; }
            acc = j_madd(acc, x, y);
        }
    }
    j_store(a.buckets + (((uint64_t)w << a.c) + b) * 24, acc);
}
// window sum  sum_b b * B_b : lane t takes buckets [t * chunk, (t + 1) * chunk) from the top down with a running sum R and an
// accumulator A (A = sum (b - lo + 1) B_b), adds (lo - 1) * R by double-and-add; lane 0 of the window then adds the partials
__global__ void __launch_bounds__(256) msm_window_kernel(MsmArgs a) {
    const uint32_t w = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nb = 1u << a.c, n_chunks = (nb + a.chunk - 1) / a.chunk;
    if (t >= n_chunks) return;
    const uint32_t lo = t * a.chunk, hi = min(nb, lo + a.chunk);
    jac run = j_identity(), acc = j_identity();
    for (uint32_t b = hi; b-- > lo;) {
        run = j_add(run, j_load(a.buckets + (((uint64_t)w << a.c) + b) * 24));
        acc = j_add(acc, run);
    }
    // acc = sum_{b in chunk} (b - lo + 1) B_b  ->  + (lo - 1) * run   (lo = 0: acc counts bucket 0, the identity, once: harmless;
    // and (0 - 1) * run must then be SUBTRACTED: handle lo = 0 by removing `run` once instead)
    if (lo == 0) {
        jac neg = run;
        neg.y = m_sub<F_Q>(u_zero(), run.y);
        acc = j_add(acc, neg);
    } else {
        jac add = j_identity();
        const uint32_t k = lo - 1;
        for (int bit = 31; bit >= 0; bit--) {
            add = j_double(add);
            if ((k >> bit) & 1) add = j_add(add, run);
        }
        acc = j_add(acc, add);
    }
    j_store(a.partial + ((uint64_t)w * n_chunks + t) * 24, acc);
}
__global__ void msm_window_sum_kernel(MsmArgs a) {
    const uint32_t w = blockIdx.x;
    if (threadIdx.x) return;
    const uint32_t nb = 1u << a.c, n_chunks = (nb + a.chunk - 1) / a.chunk;
    jac s = j_identity();
    for (uint32_t t = 0; t < n_chunks; t++) s = j_add(s, j_load(a.partial + ((uint64_t)w * n_chunks + t) * 24));
    j_store(a.wsum + (uint64_t)w * 24, s);
}
// result = sum_w 2^(c w) W_w (Horner from the top window), to affine, out of Montgomery form
__global__ void msm_final_kernel(MsmArgs a) {
    if (threadIdx.x || blockIdx.x) return;
    jac r = j_identity();
    for (uint32_t w = a.n_windows; w-- > 0;) {
        for (uint32_t k = 0; k < a.c; k++) r = j_double(r);
        r = j_add(r, j_load(a.wsum + (uint64_t)w * 24));
    }
    if (j_is_identity(r)) {
        for (int i = 0; i < 8; i++) a.result[i] = 0;
        return;
    }
    const u256 zi = m_inv<F_Q>(r.z), zi2 = m_mul<F_Q>(zi, zi);
    store256(a.result, m_to_int<F_Q>(m_mul<F_Q>(r.x, zi2)));
    store256(a.result + 4, m_to_int<F_Q>(m_mul<F_Q>(r.y, m_mul<F_Q>(zi2, zi))));
}

}  // namespace gl355

using namespace gl355;

// host-side Fr helpers for the few constants a call needs (omega_n, n^-1): plain 256-bit integers with __int128
namespace {
typedef unsigned __int128 u128;
struct H256 { uint64_t l[4]; };
const uint64_t HR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};   // r
bool h_geq(const H256& a) {
    for (int i = 3; i >= 0; i--) { if (a.l[i] > HR[i]) return true; if (a.l[i] < HR[i]) return false; }
    return true;
}
H256 h_addmod(const H256& a, const H256& b) {
    H256 r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || h_geq(r)) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r.l[i] - HR[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return r;
}
H256 h_mulmod(const H256& a, const H256& b) {          // double-and-add: called a few dozen times per API call
    H256 r = {{0, 0, 0, 0}};
    for (int i = 255; i >= 0; i--) {
        r = h_addmod(r, r);
        if ((b.l[i >> 6] >> (i & 63)) & 1) r = h_addmod(r, a);
    }
    return r;
}
H256 h_powmod(H256 a, const H256& e) {
    H256 r = {{1, 0, 0, 0}};
    for (int i = 255; i >= 0; i--) {
        r = h_mulmod(r, r);
        if ((e.l[i >> 6] >> (i & 63)) & 1) r = h_mulmod(r, a);
    }
    return r;
}
u256 to_u256(const H256& a) {
    u256 r;
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)a.l[i]; r.l[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
    return r;
}
}  // namespace

extern "C" {

int32_t gl355_bn254_fr_ntt(gl355_ctx* h, uint64_t* data, uint32_t log_n, int32_t inverse) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!data) return ctx->fail(GL355_E_INVALID_ARG, "bn254_fr_ntt: null data");
    if (log_n > 26) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_fr_ntt: log_n > 26 unsupported (Fr has 2-adicity 28)");
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) return GL355_OK;
    // omega_n = ROOT^(2^(28 - log_n)) (or its inverse), as a plain integer, then to Montgomery form: * R mod r
    H256 w = inverse ? H256{{BN254C_FR_ROOT_INV_64[0], BN254C_FR_ROOT_INV_64[1], BN254C_FR_ROOT_INV_64[2], BN254C_FR_ROOT_INV_64[3]}}
                     : H256{{BN254C_FR_ROOT_64[0], BN254C_FR_ROOT_64[1], BN254C_FR_ROOT_64[2], BN254C_FR_ROOT_64[3]}};
    for (uint32_t k = log_n; k < BN254C_FR_S; k++) w = h_mulmod(w, w);
    const H256 Rm = {{BN254C_FR_ONE_64[0], BN254C_FR_ONE_64[1], BN254C_FR_ONE_64[2], BN254C_FR_ONE_64[3]}};      // R mod r
    const u256 w_mont = to_u256(h_mulmod(w, Rm));
    u256 scale = to_u256(H256{{0, 0, 0, 0}});
    if (inverse) {
        H256 e = {{HR[0] - 2, HR[1], HR[2], HR[3]}};
        scale = to_u256(h_powmod(H256{{n, 0, 0, 0}}, e));                  // n^-1 mod r, plain
    }
    Staged sd(ctx);
    GL355_TRY(sd.open(data, n * 32, 3));
    Scratch tw(ctx);
    GL355_TRY(tw.get((n / 2) * 32 + 32));
    uint64_t* d = sd.as<uint64_t>();
    const uint32_t blk = (uint32_t)((n + 255) / 256), hblk = (uint32_t)((n / 2 + 255) / 256);
    {
        ProfScope ps(ctx, "bn254_fr_ntt", n * 64);
        hipLaunchKernelGGL(fr_twiddle_kernel, dim3(hblk ? hblk : 1), dim3(256), 0, ctx->stream, tw.as<uint64_t>(), n / 2, w_mont);
        hipLaunchKernelGGL(fr_to_mont_kernel, dim3(blk), dim3(256), 0, ctx->stream, d, n);
        hipLaunchKernelGGL(fr_bitrev_kernel, dim3(blk), dim3(256), 0, ctx->stream, d, log_n);
        for (uint32_t s = 1; s <= log_n; s++)
            hipLaunchKernelGGL(fr_stage_kernel, dim3(hblk ? hblk : 1), dim3(256), 0, ctx->stream, d, tw.as<uint64_t>(), log_n, s);
        hipLaunchKernelGGL(fr_from_mont_kernel, dim3(blk), dim3(256), 0, ctx->stream, d, n, scale, inverse ? 1 : 0);
        GL355_HIP(ctx, hipGetLastError());
    }
    return sd.finish();
}

int32_t gl355_bn254_g1_msm(gl355_ctx* h, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint64_t result[8]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!result || ((!points || !scalars) && n)) return ctx->fail(GL355_E_INVALID_ARG, "bn254_g1_msm: null argument");
    if (n > (1ull << 26)) return ctx->fail(GL355_E_UNSUPPORTED, "bn254_g1_msm: more than 2^26 points");
    if (n == 0) { memset(result, 0, 64); return GL355_OK; }
    uint32_t lg = 0;
    while ((1ull << lg) < n) lg++;
    MsmArgs a;
    memset(&a, 0, sizeof a);
    a.n = n;
    a.c = lg <= 6 ? 4 : (lg - 2 > 16 ? 16 : lg - 2);             // window bits: ~4 points per bucket, at most 2^16 buckets
    a.n_windows = (256 + a.c - 1) / a.c;
    const uint64_t nb = 1ull << a.c;
    a.chunk = nb >= 4096 ? (uint32_t)(nb / 256) : 16;           // up to 256 lanes per window in the reduction
    const uint32_t n_chunks = (uint32_t)((nb + a.chunk - 1) / a.chunk);
    Staged sp(ctx), ss(ctx);
    GL355_TRY(sp.open(points, n * 64, 1));
    GL355_TRY(ss.open(scalars, n * 32, 1));
    a.points = sp.as<uint64_t>(); a.scalars = ss.as<uint64_t>();
    Scratch buf(ctx);
    const uint64_t W = a.n_windows;
    const uint64_t words32 = n * 16 + 2 * W * nb + W * n + W * nb * 24 + W * n_chunks * 24 + W * 24 + 16;
    GL355_TRY(buf.get(words32 * 4 + 64));
    uint32_t* p = buf.as<uint32_t>();
    a.pm = p; p += n * 16;
    a.hist = p; p += W * nb;
    a.cursor = p; p += W * nb;
    a.idx = p; p += W * n;
    a.buckets = p; p += W * nb * 24;
    a.partial = p; p += W * n_chunks * 24;
    a.wsum = p; p += W * 24;
    a.result = reinterpret_cast<uint64_t*>(p);
    GL355_HIP(ctx, hipMemsetAsync(a.hist, 0, W * nb * 4, ctx->stream));
    const uint32_t blk = (uint32_t)((n + 255) / 256);
    {
        ProfScope ps(ctx, "bn254_g1_msm", n * 96);
        hipLaunchKernelGGL(msm_prepare_kernel, dim3(blk), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(msm_scan_kernel, dim3((uint32_t)W), dim3(1024), 0, ctx->stream, a);
        hipLaunchKernelGGL(msm_scatter_kernel, dim3(blk), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(msm_bucket_kernel, dim3((uint32_t)((nb + 127) / 128), (uint32_t)W), dim3(128), 0, ctx->stream, a);
        hipLaunchKernelGGL(msm_window_kernel, dim3((n_chunks + 255) / 256, (uint32_t)W), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(msm_window_sum_kernel, dim3((uint32_t)W), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL(msm_final_kernel, dim3(1), dim3(64), 0, ctx->stream, a);
        GL355_HIP(ctx, hipGetLastError());
    }
    if (ptr_is_device(result)) GL355_HIP(ctx, hipMemcpyAsync(result, a.result, 64, hipMemcpyDeviceToDevice, ctx->stream));
    else GL355_HIP(ctx, ctx->d2h(result, a.result, 64));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

}  // extern "C"
