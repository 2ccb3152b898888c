// 256-bit modular sums and differences for the BN254 fields as written-out carry chains (device only).  Shared by bn254_field.cuh (m_add /
// m_sub over both fields) and bn254.cuh (the BN254-Poseidon hasher).  Operands and results are 8 x 32-bit limbs, values < 2m; T = 2m.
//
// Sum: 8 v_add_co / v_addc_co + the conditional subtraction of 2m as a second chain on its own carry register, interleaved with the first,
// + 8 selects (24 instructions).  Difference: a borrow chain, a lane mask from the last borrow, 8 masked constants and one more chain (25).
// The compiled C carries through 64-bit adds of zero-extended limbs: ~90 instructions each.  gfx950 wants two wait states between a VALU write
// of VCC / an SGPR and the VALU read of it as a carry; where no independent instruction fills them, an s_nop does (the hazard recogniser does
// not look inside inline asm).
#pragma once
#include "gl355_internal.h"

namespace gl355 {
#if defined(__HIP_DEVICE_COMPILE__)
struct bn_limbs { uint32_t l[8]; };
GL_DEV bn_limbs bn254_add_asm(const uint32_t (&al)[8], const uint32_t (&bl)[8], const uint32_t* T) {
    struct { const uint32_t (&l)[8]; } a{al}, b{bl};
    bn_limbs r, sum;
    uint64_t cb;
#define GL355_MADD_STEP(J, ADD, SUB, CIN_A, CIN_B)                                                        \
    ADD " %[s" #J "], vcc, %[a" #J "], %[b" #J "]" CIN_A "\n\t"                                       \
    SUB " %[r" #J "], %[cb], %[s" #J "], %[t" #J "]" CIN_B "\n\ts_nop 0\n\t"
    asm(GL355_MADD_STEP(0, "v_add_co_u32_e32", "v_sub_co_u32_e64", "", "")
        GL355_MADD_STEP(1, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        GL355_MADD_STEP(2, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        GL355_MADD_STEP(3, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        GL355_MADD_STEP(4, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        GL355_MADD_STEP(5, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        GL355_MADD_STEP(6, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        GL355_MADD_STEP(7, "v_addc_co_u32_e32", "v_subb_co_u32_e64", ", vcc", ", %[cb]")
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %[r0], %[r0], %[s0], %[cb]\n\tv_cndmask_b32_e64 %[r1], %[r1], %[s1], %[cb]\n\t"
        "v_cndmask_b32_e64 %[r2], %[r2], %[s2], %[cb]\n\tv_cndmask_b32_e64 %[r3], %[r3], %[s3], %[cb]\n\t"
        "v_cndmask_b32_e64 %[r4], %[r4], %[s4], %[cb]\n\tv_cndmask_b32_e64 %[r5], %[r5], %[s5], %[cb]\n\t"
        "v_cndmask_b32_e64 %[r6], %[r6], %[s6], %[cb]\n\tv_cndmask_b32_e64 %[r7], %[r7], %[s7], %[cb]"
        : [r0] "=&v"(r.l[0]), [r1] "=&v"(r.l[1]), [r2] "=&v"(r.l[2]), [r3] "=&v"(r.l[3]), [r4] "=&v"(r.l[4]), [r5] "=&v"(r.l[5]),
          [r6] "=&v"(r.l[6]), [r7] "=&v"(r.l[7]), [s0] "=&v"(sum.l[0]), [s1] "=&v"(sum.l[1]), [s2] "=&v"(sum.l[2]), [s3] "=&v"(sum.l[3]),
          [s4] "=&v"(sum.l[4]), [s5] "=&v"(sum.l[5]), [s6] "=&v"(sum.l[6]), [s7] "=&v"(sum.l[7]), [cb] "=&s"(cb)
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]), [b4] "v"(b.l[4]), [b5] "v"(b.l[5]),
          [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [t0] "v"(T[0]), [t1] "v"(T[1]), [t2] "v"(T[2]), [t3] "v"(T[3]), [t4] "v"(T[4]), [t5] "v"(T[5]),
          [t6] "v"(T[6]), [t7] "v"(T[7])
        : "vcc");
#undef GL355_MADD_STEP
    return r;
}
GL_DEV bn_limbs bn254_sub_asm(const uint32_t (&al)[8], const uint32_t (&bl)[8], const uint32_t* tm) {
    struct { const uint32_t (&l)[8]; } a{al}, b{bl};
    // a - b, plus 2m where that borrowed: a representative in [0, 2m) as the C form gives (not always the same one; every consumer
    // either multiplies on or canonicalises)
    bn_limbs d, t;
    uint32_t mk;
    asm("v_sub_co_u32_e32 %[d0], vcc, %[a0], %[b0]\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d1], vcc, %[a1], %[b1], vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d2], vcc, %[a2], %[b2], vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d3], vcc, %[a3], %[b3], vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d4], vcc, %[a4], %[b4], vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d5], vcc, %[a5], %[b5], vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d6], vcc, %[a6], %[b6], vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %[d7], vcc, %[a7], %[b7], vcc\n\ts_nop 1\n\t"
        "v_cndmask_b32_e64 %[mk], 0, -1, vcc\n\t"
        "v_and_b32_e32 %[t0], %[m0], %[mk]\n\tv_and_b32_e32 %[t1], %[m1], %[mk]\n\t"
        "v_add_co_u32_e32 %[d0], vcc, %[d0], %[t0]\n\t"
        "v_and_b32_e32 %[t2], %[m2], %[mk]\n\tv_and_b32_e32 %[t3], %[m3], %[mk]\n\t"
        "v_addc_co_u32_e32 %[d1], vcc, %[d1], %[t1], vcc\n\t"
        "v_and_b32_e32 %[t4], %[m4], %[mk]\n\tv_and_b32_e32 %[t5], %[m5], %[mk]\n\t"
        "v_addc_co_u32_e32 %[d2], vcc, %[d2], %[t2], vcc\n\t"
        "v_and_b32_e32 %[t6], %[m6], %[mk]\n\tv_and_b32_e32 %[t7], %[m7], %[mk]\n\t"
        "v_addc_co_u32_e32 %[d3], vcc, %[d3], %[t3], vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %[d4], vcc, %[d4], %[t4], vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %[d5], vcc, %[d5], %[t5], vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %[d6], vcc, %[d6], %[t6], vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %[d7], vcc, %[d7], %[t7], vcc"
        : [d0] "=&v"(d.l[0]), [d1] "=&v"(d.l[1]), [d2] "=&v"(d.l[2]), [d3] "=&v"(d.l[3]), [d4] "=&v"(d.l[4]), [d5] "=&v"(d.l[5]),
          [d6] "=&v"(d.l[6]), [d7] "=&v"(d.l[7]), [t0] "=&v"(t.l[0]), [t1] "=&v"(t.l[1]), [t2] "=&v"(t.l[2]), [t3] "=&v"(t.l[3]),
          [t4] "=&v"(t.l[4]), [t5] "=&v"(t.l[5]), [t6] "=&v"(t.l[6]), [t7] "=&v"(t.l[7]), [mk] "=&v"(mk)
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]), [b4] "v"(b.l[4]), [b5] "v"(b.l[5]),
          [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [m0] "s"(tm[0]), [m1] "s"(tm[1]), [m2] "s"(tm[2]), [m3] "s"(tm[3]), [m4] "s"(tm[4]),
          [m5] "s"(tm[5]), [m6] "s"(tm[6]), [m7] "s"(tm[7])
        : "vcc");
    return d;
}
#endif
}  // namespace gl355
