#!/usr/bin/env python3
"""Kernel-level timing of the hot path on one GPU (HIP events via gl355_profile_*).
  python tools/kbench.py [lde|ntt|commit|merkle|poseidon|all]
Prints algorithmic GB/s and, for hash kernels, permutations/s."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gl = importlib.import_module("stark-verifier_amd")


def dev_rand(shape, seed=1):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(0, (1 << 63) - 1, shape, dtype=torch.int64, device="cuda", generator=g)


# KBENCH_WARM / KBENCH_REPS_X: untimed warm-up calls and a multiplier on every repetition count -- the first ~10 launches after the device
# did something else run 15-20 % slow (profiles/r03_ubench_ntt_l24s.txt); steady state: KBENCH_WARM=20 KBENCH_REPS_X=10
WARM = int(os.environ.get("KBENCH_WARM", "1"))
REPS_X = int(os.environ.get("KBENCH_REPS_X", "1"))


def timed(ctx, fn, reps=5, warm=None):
    warm = WARM if warm is None else warm
    reps *= REPS_X
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_read()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    ms = ctx.timer_stop() / reps
    prof = {k: v[1] / reps for k, v in ctx.profile_read().items() if not k.startswith("host:")}
    ctx.profile_enable(False)
    return ms, prof


def bench_lde(ctx, log_n, rate_bits, batch, reps=5):
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    c = dev_rand((batch, n))
    out = torch.empty((batch, N), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ms, prof = timed(ctx, lambda: ctx.check(ctx.lib.gl355_lde_bitrev(ctx.h, C.c_void_p(c.data_ptr()), log_n, rate_bits, 7, batch,
                                                                    C.c_void_p(out.data_ptr()))), reps)
    alg = 8.0 * batch * (n + N)
    print("lde    n=2^%-2d N=2^%-2d B=%-4d %8.3f ms  %8.1f GB/s alg  %s" % (log_n, log_n + rate_bits, batch, ms, alg / ms / 1e6,
                                                                        {k: round(v, 3) for k, v in prof.items()}))


def bench_ntt(ctx, log_n, batch, inverse=0, reps=5):
    n = 1 << log_n
    x = dev_rand((batch, n))
    torch.cuda.synchronize()
    ms, prof = timed(ctx, lambda: ctx.check(ctx.lib.gl355_ntt(ctx.h, C.c_void_p(x.data_ptr()), log_n, batch, n, inverse)), reps)
    alg = 16.0 * batch * n
    print("ntt%s n=2^%-2d        B=%-4d %8.3f ms  %8.1f GB/s alg  %s" % ("-inv" if inverse else "    ", log_n, batch, ms, alg / ms / 1e6,
                                                                       {k: round(v, 3) for k, v in prof.items()}))


def bench_merkle(ctx, log_n, leaf_len, cap, reps=3):
    n = 1 << log_n
    leaves = dev_rand((n, leaf_len))
    dig = torch.empty((2 * (n - (1 << cap)), 4), dtype=torch.int64, device="cuda")
    capb = torch.empty((1 << cap, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ms, prof = timed(ctx, lambda: ctx.check(ctx.lib.gl355_merkle_build(ctx.h, C.c_void_p(leaves.data_ptr()), n, leaf_len, cap,
                                                                      C.c_void_p(dig.data_ptr()), C.c_void_p(capb.data_ptr()))), reps)
    perms = n * ((leaf_len + 7) // 8 if leaf_len > 4 else 0) + (n - (1 << cap))
    alg = 8.0 * n * leaf_len + 64.0 * (n - (1 << cap)) + 32.0 * (1 << cap)
    print("merkle N=2^%-2d L=%-4d cap=%d   %8.3f ms  %8.1f Mperm/s  %7.1f GB/s alg  %s" % (log_n, leaf_len, cap, ms, perms / ms / 1e3,
                                                                                        alg / ms / 1e6, {k: round(v, 3) for k, v in prof.items()}))


def bench_poseidon(ctx, log_n, reps=3):
    n = 1 << log_n
    st = dev_rand((n, 12))
    torch.cuda.synchronize()
    ms, _ = timed(ctx, lambda: ctx.check(ctx.lib.gl355_poseidon_permute(ctx.h, C.c_void_p(st.data_ptr()), n)), reps)
    print("poseidon_permute 2^%-2d         %8.3f ms  %8.1f Mperm/s" % (log_n, ms, n / ms / 1e3))


def bench_bn254(ctx, log_n, reps=3):
    n = 1 << log_n
    st = dev_rand((n, 12))
    torch.cuda.synchronize()
    ms, _ = timed(ctx, lambda: ctx.check(ctx.lib.gl355_permute_h(ctx.h, 1, C.c_void_p(st.data_ptr()), n)), reps)
    print("bn254_permute 2^%-2d            %8.3f ms  %8.2f Mperm/s" % (log_n, ms, n / ms / 1e3))


def bench_merkle_bn254(ctx, log_n, leaf_len, cap, reps=2):
    n = 1 << log_n
    leaves = dev_rand((n, leaf_len))
    dig = torch.empty((2 * (n - (1 << cap)), 4), dtype=torch.int64, device="cuda")
    capb = torch.empty((1 << cap, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ms, prof = timed(ctx, lambda: ctx.check(ctx.lib.gl355_merkle_build_h(ctx.h, 1, C.c_void_p(leaves.data_ptr()), n, leaf_len, cap,
                                                                        C.c_void_p(dig.data_ptr()), C.c_void_p(capb.data_ptr()))), reps)
    perms = n * ((leaf_len + 7) // 8 if leaf_len > 4 else 0) + (n - (1 << cap))
    print("merkle-bn254 N=2^%-2d L=%-4d cap=%d   %8.3f ms  %8.2f Mperm/s  %s" % (log_n, leaf_len, cap, ms, perms / ms / 1e3,
                                                                              {k: round(v, 3) for k, v in prof.items()}))


def bench_commit(ctx, log_n, batch, rate_bits=3, cap=4, salted=True, reps=3):
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    vals = dev_rand((batch, n))
    salt = dev_rand((4, N), 2) if salted else None
    torch.cuda.synchronize()

    def run():
        h = C.c_void_p()
        ctx.check(ctx.lib.gl355_commit(ctx.h, C.c_void_p(vals.data_ptr()), log_n, batch, rate_bits, 0,
                                       C.c_void_p(salt.data_ptr()) if salted else None, cap, C.byref(h)))
        ctx.lib.gl355_oracle_destroy(h)
    ms, prof = timed(ctx, run, reps)
    print("commit n=2^%-2d B=%-4d salted=%d  %8.3f ms  %s" % (log_n, batch, salted, ms, {k: round(v, 3) for k, v in prof.items()}))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ctx = gl.Context(0)
    if what == "lde17":     # the cfg-2 (B) shape alone, short and long: the device's clock settles over tens of milliseconds
        for reps in (5, 50, 200):
            bench_lde(ctx, 17, 3, 135, reps=reps)
    if what in ("lde", "all"):
        bench_lde(ctx, 17, 3, 135)
        bench_lde(ctx, 17, 3, 16)
        bench_lde(ctx, 13, 3, 135)
        bench_lde(ctx, 14, 3, 135)
        bench_lde(ctx, 15, 3, 135)
        bench_lde(ctx, 12, 3, 135)
    if what == "nttbig":    # cfg-2 (C): forward NTT, batch 16, 2^19 .. 2^23 points (bit-reversed output, the two-pass / three-pass boundary)
        for ln in (19, 20, 21, 22, 23):
            bench_ntt(ctx, ln, 16, reps=10)
    if what in ("ntt", "all"):
        for b in (1, 16, 135):
            bench_ntt(ctx, 20, b)
        bench_ntt(ctx, 13, 135, inverse=1)
        bench_ntt(ctx, 16, 135)
    if what in ("poseidon", "all"):
        bench_poseidon(ctx, 22)
    if what in ("merkle", "all"):
        bench_merkle(ctx, 22, 4, 4)
        bench_merkle(ctx, 20, 4, 0)
        bench_merkle(ctx, 20, 135, 4)
        bench_merkle(ctx, 16, 139, 4)
    if what in ("bn254", "all"):
        bench_bn254(ctx, 20)
        bench_bn254(ctx, 16)
        bench_merkle_bn254(ctx, 17, 139, 4)
        bench_merkle_bn254(ctx, 16, 4, 4)
    if what in ("commit", "all"):
        bench_commit(ctx, 13, 135)
        bench_commit(ctx, 13, 85, salted=False)
        bench_commit(ctx, 13, 20)
        bench_commit(ctx, 15, 135)
    ctx.close()


if __name__ == "__main__":
    main()
