#!/usr/bin/env python3
"""profiles/<round>_lde_pmc.json from the LDE summaries tools/prof_round5.sh leaves in gpurun_out/<dir>/ (ldepmc_FETCH_SIZE / _WRITE_SIZE / _sq:
separate rocprofv3 --pmc passes over tools/prof_lde.py, per-dispatch averages).  FETCH_SIZE is doubled per the gfx950 note of
MI355X_MICROARCH.md (HBM section), WRITE_SIZE taken as reported; both are in KB.
usage: make_lde_pmc.py gpurun_out/prof_r05 profiles/r05_lde_pmc.json"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
N_OUT = 135 << 20          # output elements of the bench shape


def read(name):
    out = {}
    for line in open("%s/%s_summary.txt" % (src, name)):
        kern, rest = line.rsplit(": ", 1)
        out[kern] = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)=([\d.e+]+) \(n=\d+\)", rest)}
    return out


fetch, write, sq = read("ldepmc_FETCH_SIZE"), read("ldepmc_WRITE_SIZE"), read("ldepmc_sq")
kernels, total_b, total_i = {}, 0, 0.0
for kern, s in sq.items():
    if "ntt_" not in kern:
        continue
    name = re.sub(r"^void gl355::", "", kern)
    name = re.sub(r"\(.*$", "", name)
    f, w = fetch[kern]["FETCH_SIZE"], write[kern]["WRITE_SIZE"]
    b = int((2 * f + w) * 1024)
    kernels[name] = {"FETCH_SIZE_KB_reported": f, "WRITE_SIZE_KB": w, "SQ_INSTS_VALU": s["SQ_INSTS_VALU"], "SQ_BUSY_CYCLES": s["SQ_BUSY_CYCLES"],
                     "SQ_WAIT_INST_ANY": s.get("SQ_WAIT_INST_ANY"), "SQ_WAVES": s.get("SQ_WAVES"), "SQ_INSTS_VALU_INT64": s.get("SQ_INSTS_VALU_INT64"),
                     "SQ_INSTS_VALU_INT32": s.get("SQ_INSTS_VALU_INT32"), "hbm_bytes_per_launch": b,
                     "clk_per_valu_inst_per_simd": round((s["SQ_BUSY_CYCLES"] / 32) / (s["SQ_INSTS_VALU"] / 1024), 2),
                     "lane_insts_per_output_element": round(s["SQ_INSTS_VALU"] * 64 / N_OUT, 1)}
    total_b += b
    total_i += s["SQ_INSTS_VALU"]
doc = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU (separate passes over "
                  "tools/prof_lde.py; tools/prof_round6.sh lde, tools/make_lde_pmc.py); FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md",
       "workload": "lde 2^17 -> 2^20 x 135 columns, bit-reversed output (split-exchange 24-bit-limb column pass + split-exchange 24-bit-limb row pass)",
       "kernels": kernels, "hbm_bytes_per_lde": total_b, "valu_insts_per_lde": total_i}
json.dump(doc, open(dst, "w"), indent=1)
print("wrote", dst, total_b, total_i)
