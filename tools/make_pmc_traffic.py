#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from the per-kernel summaries tools/prof_round1c.sh leaves in gpurun_out/<dir>/:
HBM-side bytes per launch (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as reported; both in KB)
and cycles per VALU instruction per SIMD (SQ_INSTS_VALU / 1024 SIMDs against SQ_BUSY_CYCLES / 32 shader engines).
usage: make_pmc_traffic.py gpurun_out/prof_r01c profiles/r01_pmc_traffic.json [bench line of the SQ pass]
With the bench line of the SQ pass (its config.units_proven_in_process) the file also gets job.valu_insts_per_unit: the wave-level VALU
instructions of every kernel of that process over the units it proved."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]


def read(name):
    out = {}
    for line in open("%s/%s_summary.txt" % (src, name)):
        kern, rest = line.rsplit(": ", 1)
        vals = {m.group(1): (float(m.group(2)), int(m.group(3))) for m in re.finditer(r"(\w+)=([\d.e+]+) \(n=(\d+)\)", rest)}
        out[kern] = vals
    return out


def short(kern):
    k = re.sub(r"^void ", "", kern)
    k = re.sub(r"^gl355::", "", k)
    return re.sub(r"\(.*$", "", k)


fetch, write, sq = read("pmc_FETCH_SIZE"), read("pmc_WRITE_SIZE"), read("pmc_sq")
kernels = {}
for kern, v in fetch.items():
    f, n = v["FETCH_SIZE"]
    w = write.get(kern, {}).get("WRITE_SIZE", (0.0, 0))[0]
    e = {"fetch_kb_reported": f, "write_kb_reported": w, "launches": n, "bytes_per_launch_corrected": int((2 * f + w) * 1024)}
    s = sq.get(kern)
    if s and "SQ_INSTS_VALU" in s and s["SQ_BUSY_CYCLES"][0] > 0:
        e["valu_insts_per_launch"] = s["SQ_INSTS_VALU"][0]
        e["busy_cycles_per_se"] = s["SQ_BUSY_CYCLES"][0] / 32
        e["clk_per_valu_inst_per_simd"] = round((s["SQ_BUSY_CYCLES"][0] / 32) / (s["SQ_INSTS_VALU"][0] / 1024), 2) if s["SQ_INSTS_VALU"][0] else None
        e["sq_launches"] = s["SQ_INSTS_VALU"][1]
        # dynamic class counts where the pass carried them (round 4 on): SQ_INSTS_VALU_INT64 = v_mad_u64_u32 and the 64-bit shifts / adds
        # (calibrated against the probe kernels of csrc/valu_probe.hip in the same pass: profiles/rNN_valu_probe_pmc.txt)
        if "SQ_INSTS_VALU_INT64" in s:
            e["valu_int64_per_launch"] = s["SQ_INSTS_VALU_INT64"][0]
        if "SQ_INSTS_VALU_INT32" in s:
            e["valu_int32_per_launch"] = s["SQ_INSTS_VALU_INT32"][0]
    kernels[short(kern)] = e
doc = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python bench.py --steps 1 --warmup 0 --proofs-per-step 16 "
                  "--threads 1 --no-cpu-baseline` (one context, lock-step batches of 8 units; tools/prof_round5.sh, tools/make_pmc_traffic.py), per-dispatch averages in KB; FETCH_SIZE doubled "
                  "per the gfx950 note of MI355X_MICROARCH.md (HBM section); WRITE_SIZE taken as reported",
       "_valu_note": "SQ_INSTS_VALU (wave instructions, whole chip) / 1024 SIMDs vs SQ_BUSY_CYCLES / 32 shader engines, same rocprofv3 --pmc run: "
                     "cycles per VALU instruction per SIMD; the issue floor measured by tools/ubench is ~4.2-4.4",
       "kernels": kernels}
if len(sys.argv) > 3:
    try:
        line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        n_units = int(line["config"]["units_proven_in_process"])
        # (the probe kernels of gl355_valu_probe run in the same process: not part of a unit)
        total = sum(v["SQ_INSTS_VALU"][0] * v["SQ_INSTS_VALU"][1] for k, v in sq.items() if "SQ_INSTS_VALU" in v and not short(k).startswith("vp_"))
        doc["job"] = {"units_proven_in_process": n_units, "valu_insts_total": total, "valu_insts_per_unit": round(total / n_units)}
    except Exception as exc:
        print("no job figure:", exc)
json.dump(doc, open(dst, "w"), indent=1)
print("wrote", dst, len(kernels), "kernels")
