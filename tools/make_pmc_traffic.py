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


import os
fetch, write = read("pmc_FETCH_SIZE"), read("pmc_WRITE_SIZE")
# round 6: two SQ passes of the same one-context job, 1 step and 3 steps; their difference is two steps' kernels WITHOUT the set-up (key hashes, group
# tree, preprocessed commitments of the two circuits), VERDICT r5 #1 (c).  Older layouts have the single pass pmc_sq.
steady = os.path.exists("%s/pmc_sq3_summary.txt" % src) and os.path.exists("%s/pmc_sq1_summary.txt" % src)
sq = read("pmc_sq3") if steady else read("pmc_sq")
sq1 = read("pmc_sq1") if steady else {}
kernels = {}
for kern, v in fetch.items():
    f, n = v["FETCH_SIZE"]
    w = write.get(kern, {}).get("WRITE_SIZE", (0.0, 0))[0]
    e = {"fetch_kb_reported": f, "write_kb_reported": w, "launches": n, "bytes_per_launch_corrected": int((2 * f + w) * 1024)}
    s = sq.get(kern)
    if s and "SQ_INSTS_VALU" in s and s.get("SQ_BUSY_CYCLES", (0, 0))[0] > 0:
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
if steady:
    for kern, v3 in sq.items():
        v1 = sq1.get(kern, {})
        e = kernels.setdefault(short(kern), {})
        for ctr, key in (("SQ_INSTS_VALU", "steady_valu_insts"), ("SQ_INSTS_VALU_INT64", "steady_valu_int64"), ("SQ_INSTS_VALU_INT32", "steady_valu_int32")):
            if ctr in v3:
                t3 = v3[ctr][0] * v3[ctr][1]
                t1 = v1[ctr][0] * v1[ctr][1] if ctr in v1 else 0.0
                e[key] = t3 - t1
        if "SQ_INSTS_VALU" in v3:
            e["steady_launches"] = v3["SQ_INSTS_VALU"][1] - (v1["SQ_INSTS_VALU"][1] if "SQ_INSTS_VALU" in v1 else 0)
doc = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python bench.py --steps 1 --warmup 0 --proofs-per-step 16 "
                  "--threads 1 --no-cpu-baseline` (one context, lock-step batches of 8 units; tools/prof_round5.sh, tools/make_pmc_traffic.py), per-dispatch averages in KB; FETCH_SIZE doubled "
                  "per the gfx950 note of MI355X_MICROARCH.md (HBM section); WRITE_SIZE taken as reported",
       "_valu_note": "SQ_INSTS_VALU (wave instructions, whole chip) / 1024 SIMDs vs SQ_BUSY_CYCLES / 32 shader engines, same rocprofv3 --pmc run: "
                     "cycles per VALU instruction per SIMD; the issue floor measured by tools/ubench is ~4.2-4.4",
       "kernels": kernels}
if steady and len(sys.argv) > 4:
    # argv[3] / argv[4]: the bench lines of the 1-step and the 3-step SQ pass
    l1 = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    l3 = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
    u1, u3 = int(l1["config"]["units_proven_in_process"]), int(l3["config"]["units_proven_in_process"])
    tot = {"valu": 0.0, "int64": 0.0, "int32": 0.0}
    incl = 0.0
    for k, e in kernels.items():
        if k.startswith("vp") or "steady_valu_insts" not in e:
            continue
        tot["valu"] += e["steady_valu_insts"]
        tot["int64"] += e.get("steady_valu_int64", 0.0)
        tot["int32"] += e.get("steady_valu_int32", 0.0)
    for kern, v in sq.items():
        if "SQ_INSTS_VALU" in v and not short(kern).startswith("vp"):
            incl += v["SQ_INSTS_VALU"][0] * v["SQ_INSTS_VALU"][1]
    n = u3 - u1
    for k, e in kernels.items():
        if "steady_valu_insts" in e:
            e["steady_valu_insts_per_unit"] = round(e["steady_valu_insts"] / n)
    doc["job"] = {"how": "SQ_INSTS_VALU of every kernel: (3-step pass) - (1-step pass) of the same one-context job, over the units proven in between; the "
                         "set-up kernels (2^20 key hashes, the group tree, two circuits' preprocessed commitments) cancel",
                  "units_steady": n, "units_in_3_step_pass": u3, "valu_insts_steady_total": tot["valu"],
                  "valu_insts_per_unit": round(tot["valu"] / n), "valu_int64_per_unit": round(tot["int64"] / n), "valu_int32_per_unit": round(tot["int32"] / n),
                  "valu_insts_per_unit_including_setup": round(incl / u3)}
elif len(sys.argv) > 3:
    try:
        line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        n_units = int(line["config"]["units_proven_in_process"])
        # (the probe kernels of gl355_valu_probe run in the same process: not part of a unit)
        total = sum(v["SQ_INSTS_VALU"][0] * v["SQ_INSTS_VALU"][1] for k, v in sq.items() if "SQ_INSTS_VALU" in v and not short(k).startswith("vp_"))
        doc["job"] = {"units_proven_in_process": n_units, "valu_insts_total": total, "valu_insts_per_unit": round(total / n_units)}
    except Exception as exc:
        print("no job figure:", exc)
# the composite probes of csrc/valu_probe.hip under the same counters (pmc_probe_summary.txt): dynamic instructions per product / per permutation
if os.path.exists("%s/pmc_probe_summary.txt" % src):
    pr = read("pmc_probe")
    per_lane = {"vpc_product_kernel": 8 * 2048, "vpc_permute_kernel": 48}       # VP_PROD_ITERS x 8 products, VP_PERM_ITERS (csrc/valu_probe.hip)
    doc["probes"] = {}
    for kern, v in pr.items():
        k = short(kern)
        if k in per_lane and "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
            items = v["SQ_WAVES"][0] * per_lane[k]                               # wave-level items per launch
            doc["probes"][k] = {"valu_insts_per_item": round(v["SQ_INSTS_VALU"][0] / items, 2),
                                "valu_int64_per_item": round(v.get("SQ_INSTS_VALU_INT64", (0, 0))[0] / items, 2),
                                "valu_int32_per_item": round(v.get("SQ_INSTS_VALU_INT32", (0, 0))[0] / items, 2)}
json.dump(doc, open(dst, "w"), indent=1)
print("wrote", dst, len(kernels), "kernels")
