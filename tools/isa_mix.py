#!/usr/bin/env python3
"""Static VALU instruction-class histogram of every kernel the library ships, from the gfx950 ISA hipcc emits for csrc/*.hip
(hipcc -S --cuda-device-only): the `mix` half of the VALU roofline (bench.py: peak = 1 / sum_c f_c / rate_c with the class rates
gl355_valu_probe measures in the same run).  Classes as in csrc/valu_probe.hip / tools/ubench/ubench_alu2.hip:
  full32  add / sub / logic / right shifts / 32-bit moves
  mad64   v_mad_u64_u32 / v_mad_i64_i32 and the 64-bit shifts, adds and moves
  half32  every other VALU opcode (carry adds, left shifts, v_mul_lo, three-operand forms, v_cndmask, v_perm, lane ops ...)
usage: isa_mix.py profiles/rNN_isa_mix.json   (runs in this container: no GPU needed)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stark-verifier_amd", "csrc")
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_add_f32", "v_mul_f32", "v_sub_f32"}
MAD64 = {"v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_mov_b64", "v_pk_mov_b32"}


def classify(op):
    op = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if op in FULL:
        return "full32"
    if op in MAD64:
        return "mad64"
    return "half32"


def form_of(op, args):
    """the opcode FORM of csrc/valu_probe.hip (gl355_valu_probe_op_name) an instruction is priced as; None = no probe covers it (bench.py prices
    those at the fastest measured form, so the ceiling stays a ceiling)"""
    op = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    a = [x.strip() for x in args.split(",")] if args else []
    if op in ("v_add_u32",):
        return "v_add_u32"
    if op in ("v_sub_u32", "v_subrev_u32"):
        return "v_sub_u32"
    if op in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32"):
        return "v_and_b32"
    if op in ("v_lshrrev_b32",):
        return "v_lshrrev_b32"
    if op in ("v_ashrrev_i32",):
        return "v_ashrrev_i32"
    if op in ("v_mov_b32",):
        return "v_mov_b32"
    if op in ("v_lshlrev_b32",):
        return "v_lshlrev_b32"
    if op in ("v_alignbit_b32", "v_alignbyte_b32"):
        return "v_alignbit_b32"
    if op in ("v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_lshl_or_b32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_perm_b32",
              "v_xad_u32", "v_or3_b32", "v_mad_u32_u24", "v_mad_i32_i24"):
        return "v_add3_u32"
    if op in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mul_hi_i32"):
        return "v_mul_lo_u32"
    if op in ("v_add_co_u32",):
        return "v_add_co_u32 sgpr"
    if op in ("v_sub_co_u32", "v_subrev_co_u32"):
        return "v_sub_co_u32 sgpr"
    if op in ("v_addc_co_u32",):
        return "v_addc_co_u32 sgpr"
    if op in ("v_subb_co_u32", "v_subbrev_co_u32"):
        return "v_subb_co_u32 sgpr"
    if op == "v_cndmask_b32":
        srcs = a[1:3]
        return "v_cndmask_b32 0,-1,sgpr" if all(re.fullmatch(r"-?\d+", x) for x in srcs) else "v_cndmask_b32 sgpr"
    if op in ("v_mad_u64_u32", "v_mad_i64_i32"):
        srcs = a[2:4]
        if any(re.match(r"s\d|s\[|vcc|ttmp", x) for x in srcs):
            return "v_mad_u64_u32 svv"
        if "-1" in srcs:
            return "v_mad_u64_u32 v,-1,v"
        if any(re.fullmatch(r"-?(\d+|0x[0-9a-f]+)", x) for x in srcs):
            return "v_mad_u64_u32 vcv"
        return "v_mad_u64_u32 vvv"
    if op == "v_lshl_add_u64":
        return "v_lshl_add_u64"
    if op == "v_lshlrev_b64":
        return "v_lshlrev_b64"
    if op in ("v_lshrrev_b64", "v_ashrrev_i64"):
        return "v_lshrrev_b64"
    if op in ("v_mov_b64", "v_pk_mov_b32"):
        return "v_mov_b64"
    if re.match(r"v_cmp_\w+_[ui]64$", op):
        return "v_cmp_lt_u64 sgpr"
    return None


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    dst = sys.argv[1]
    kernels = {}
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                                   os.path.join(CSRC, src), "-o", tmp.name], stderr=subprocess.DEVNULL)
            txt = open(tmp.name).read()
        # functions: kernels (.amdhsa_kernel) and the noinline device functions they call
        is_kernel = set(re.findall(r"\.amdhsa_kernel (\S+)", txt))
        found = {}
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)(?:s_endpgm|s_setpc_b64)", txt, re.S | re.M):
            name, body = m.group(1), m.group(2)
            c = collections.Counter()
            ops = collections.Counter()
            forms = collections.Counter()
            for l in body.split("\n"):
                t = l.strip()
                if not l.startswith("\t") or not t or t[0] in ";.":
                    continue
                toks = t.split(None, 1)
                op = toks[0]
                if op.startswith("v_"):
                    c[classify(op)] += 1
                    ops[re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)] += 1
                    f = form_of(op, toks[1] if len(toks) > 1 else "")
                    forms[f if f else "unprobed:" + re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)] += 1
            found[name] = (c, ops, forms)
        dm = demangle(list(found))
        for name, (c, ops, forms) in found.items():
            n = sum(c.values())
            if n == 0:
                continue
            short = re.sub(r"^void ", "", dm[name])
            short = re.sub(r"^gl355::", "", short)
            short = re.sub(r"\(.*$", "", short)
            kernels.setdefault(short, {"source": src, "kernel": name in is_kernel, "valu_static": 0, "full32": 0, "half32": 0, "mad64": 0, "top": {}, "forms": {}})
            k = kernels[short]
            k["valu_static"] += n
            for cl in ("full32", "half32", "mad64"):
                k[cl] += c[cl]
            for o, v in ops.most_common(6):
                k["top"][o] = k["top"].get(o, 0) + v
            for o, v in forms.items():
                k["forms"][o] = k["forms"].get(o, 0) + v
    for k in kernels.values():
        n = k["valu_static"]
        k["f"] = {cl: round(k[cl] / n, 4) for cl in ("full32", "half32", "mad64")}
    doc = {"_source": "tools/isa_mix.py: hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only over stark-verifier_amd/csrc/*.hip, static VALU opcode "
                      "histogram per function (template instances of one name summed), classes of csrc/valu_probe.hip; `forms` = the complete histogram "
                      "by the opcode forms gl355_valu_probe_ops measures (operand kinds of v_mad_u64_u32 / v_cndmask told apart; `unprobed:<op>` = no probe)",
           "kernels": kernels}
    json.dump(doc, open(dst, "w"), indent=1, sort_keys=True)
    for name in ("hash_leaves_kernel", "merkle_level_kernel", "quotient_kernel"):
        for k, v in kernels.items():
            if k.startswith(name):
                print(k, v["valu_static"], v["f"])
    print("wrote", dst, len(kernels), "functions")


if __name__ == "__main__":
    main()
