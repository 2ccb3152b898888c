#!/usr/bin/env python3
"""Build-time check of the gfx950 "VALU writes an SGPR / VCC -> a VALU instruction reads it" hazard in the ISA hipcc emits for the library's kernels.

Why: the field product, the MDS recombination and the lock-step groups (csrc/gl_field.cuh gl_mul_multi, csrc/poseidon.cuh psd_recombine_multi)
keep their carries in scalar pairs and rely on the ORDER of separate inline-asm statements (pinned by scheduling barriers) to put the two wait states
the hardware needs between the VALU instruction that writes a pair and the VALU instruction that reads it.  LLVM's hazard recogniser does not look
inside inline asm, so nothing in the compiler checks that distance.  This script does, on the final ISA: for every VALU instruction
that reads an SGPR or VCC, the wait states since the last VALU write of that register inside the same basic block (every issued instruction counts
one, `s_nop N` counts N + 1) must be >= 2.  Labels and branches end a block (the compiler's own recogniser covers what crosses them).

usage: check_hazards.py [file.hip ...]      (default: the translation units with hand-placed carries)   exit code 1 on a violation
       check_hazards.py --asm file.s        check an already generated ISA listing"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stark-verifier_amd", "csrc")
DEFAULT = ["merkle.hip", "quotient.hip", "fri.hip"]
NEED = 2

SREG = re.compile(r"\b(s\[\d+:\d+\]|s\d+|vcc_lo|vcc_hi|vcc)\b")


def regs_of(tok):
    """scalar registers named by one operand: s5 -> {5}; s[4:5] -> {4, 5}; vcc -> {'vcc'}"""
    tok = tok.strip()
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def sdst_index(op):
    """position of the scalar destination among the operands of a VALU instruction, or None"""
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if base in ("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32",
                "v_subbrev_co_u32", "v_div_scale_f32", "v_div_scale_f64"):
        return 1
    if base.startswith("v_cmp_") or base.startswith("v_cmpx_"):
        return 0
    if base in ("v_readlane_b32", "v_readfirstlane_b32"):
        return None         # a different hazard class (lane ops), handled by the compiler
    return None


def check_listing(txt, name):
    bad = []
    n_checked = 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)(?:s_endpgm|s_setpc_b64)", txt, re.S | re.M):
        fn, body = m.group(1), m.group(2)
        last_write = {}          # scalar register -> wait states issued since its last VALU write
        for raw in body.split("\n"):
            t = raw.strip()
            if not t or t[0] == ";":
                continue
            if not raw.startswith("\t"):          # a label: a new basic block
                last_write.clear()
                continue
            if t[0] == ".":
                continue
            t = t.split(";")[0].strip()
            toks = t.split(None, 1)
            op = toks[0]
            args = [a.strip() for a in toks[1].split(",")] if len(toks) > 1 else []
            if op.startswith("s_cbranch") or op in ("s_branch", "s_barrier", "s_setpc_b64", "s_swappc_b64"):
                last_write.clear()
                continue
            states = 1
            if op == "s_nop":
                states = int(args[0], 0) + 1
            elif op.startswith("v_"):
                di = sdst_index(op)
                writes = regs_of(args[di]) if di is not None and di < len(args) else set()
                reads = set()
                for i, a in enumerate(args):
                    if i == 0 or i == di:
                        continue           # vdst / sdst
                    reads |= regs_of(a)
                # an e32 carry form names vcc once for the destination and once for the source: both are listed explicitly in the text
                for r in reads:
                    if r in last_write:
                        n_checked += 1
                        if last_write[r] < NEED:
                            bad.append((fn, t, r, last_write[r]))
                for r in list(last_write):
                    last_write[r] += 1
                for r in writes:
                    last_write[r] = 0
                continue
            for r in list(last_write):
                last_write[r] += states
            # a scalar instruction that overwrites the register ends the dependence
            if op.startswith("s_") and args:
                for r in regs_of(args[0]):
                    last_write.pop(r, None)
    return bad, n_checked


def listing_of(src, extra=()):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", *extra,
                               os.path.join(CSRC, src), "-o", tmp.name], stderr=subprocess.DEVNULL)
        return open(tmp.name).read()


def main():
    args = sys.argv[1:]
    total_bad = 0
    if args and args[0] == "--asm":
        jobs = [(a, open(a).read()) for a in args[1:]]
    else:
        jobs = [(f, listing_of(f)) for f in (args or DEFAULT)]
    for name, txt in jobs:
        bad, n = check_listing(txt, name)
        print("%s: %d VALU reads of a VALU-written scalar checked, %d with fewer than %d wait states" % (name, n, len(bad), NEED))
        for fn, t, r, d in bad[:20]:
            print("   %s: `%s` reads %s %d wait state(s) after its VALU write" % (fn, t, "vcc" if r == "vcc" else "s%d" % r, d))
        total_bad += len(bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
