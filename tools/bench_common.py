"""Shared pieces of bench.py and tools/bench_blocks.py: profile look-ups (the committed rocprofv3 --pmc passes), the VALU roofline's mix / peak,
the shader-clock sampler, and the N > 1 exchange (open_comm).  Moved out of bench.py in round 5: no behaviour change."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_N, RATE_BITS, BATCH = 17, 3, 135
LDE_PER_STEP = 8      # --workload lde: LDEs of BATCH columns per step
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a 256-thread host with a 128-CPU
    quota runs 256 OpenMP threads ten times slower than 128)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


def latest_profile(suffix):
    """profiles/rNN<suffix> of the latest round that has one (bench.py cannot collect PMC counters itself: they come from the
    committed rocprofv3 --pmc passes)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + suffix)))
    return c[-1] if c else None


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (bench.py cannot collect PMC counters
    itself); None when no pass covers the kernel."""
    path = latest_profile("_pmc_traffic.json")
    try:
        d = json.load(open(path))
        return d["kernels"][kernel]["bytes_per_launch_corrected"], "profiles/%s: %s" % (os.path.basename(path), d["_source"])
    except Exception:
        return None, None


N_SIMD = 1024                     # 256 CUs x 4 SIMDs
VALU_CLASSES = ("full32", "half32", "mad64")


def valu_probe(ctx):
    """gl355_valu_probe (csrc/valu_probe.hip) on this device, in this run: per instruction class the chip-wide issue rate of a kernel that
    only issues that class (G wave-instructions/s), the shader clock read inside that kernel, and the cost in shader cycles per wave
    instruction per SIMD that follows from the two (no assumed frequency anywhere)."""
    rates = (C.c_double * 3)()
    mhz = (C.c_double * 3)()
    ctx.check(ctx.lib.gl355_valu_probe(ctx.h, rates, mhz))
    return {c: {"rate_ginst_s": round(rates[i], 1), "shader_mhz": round(mhz[i]),
                "clk_per_wave_inst_per_simd": round(mhz[i] * 1e6 * N_SIMD / (rates[i] * 1e9), 3) if rates[i] > 0 else None}
            for i, c in enumerate(VALU_CLASSES)}


VALU_OP_ILPS = (1, 4, 8)


def valu_probe_ops(ctx, ilps=VALU_OP_ILPS):
    """gl355_valu_probe_ops: per opcode FORM the issue cost in shader cycles per wave instruction per SIMD, at 1 / 4 / 8 independent chains per lane
    (8 waves per SIMD on every SIMD); `clk` = the best of the three = what the ceiling is priced with.  -> {form: {"clk": c, "ilp1": .., "ilp4": .., "ilp8": .., "mhz": ..}}"""
    lib = ctx.lib
    names = []
    i = 0
    while True:
        n = lib.gl355_valu_probe_op_name(i)
        if not n:
            break
        names.append(n.decode())
        i += 1
    out = {n: {} for n in names}
    for ilp in ilps:
        rates = (C.c_double * len(names))()
        mhz = (C.c_double * len(names))()
        ctx.check(lib.gl355_valu_probe_ops(ctx.h, ilp, rates, mhz))
        for k, n in enumerate(names):
            clk = mhz[k] * 1e6 * N_SIMD / (rates[k] * 1e9) if rates[k] > 0 else None
            out[n]["ilp%d" % ilp] = round(clk, 3) if clk else None
            if clk and (out[n].get("clk") is None or clk < out[n]["clk"]):
                out[n]["clk"] = round(clk, 3)
                out[n]["mhz"] = round(mhz[k])
    return out


def valu_probe_composites(ctx):
    """gl355_valu_probe_composite: the shipped product / permutation on register operands, and the two mixed-class experiments.
    -> {name: {"items_g_per_s", "mhz", "waves_per_simd", "simd_clk_per_wave_item"}}"""
    out = {}
    w = -1
    while True:
        w += 1
        name = ctx.lib.gl355_valu_probe_composite_name(w)
        if not name:
            break
        name = name.decode()
        r, m, wv = C.c_double(0), C.c_double(0), C.c_uint32(0)
        ctx.check(ctx.lib.gl355_valu_probe_composite(ctx.h, w, C.byref(r), C.byref(m), C.byref(wv)))
        wave_items = r.value * 1e9 / 64                      # wave-level items per second, whole chip
        out[name] = {"items_g_per_s": round(r.value, 3), "mhz": round(m.value), "waves_per_simd": wv.value,
                     "simd_clk_per_wave_item": round(N_SIMD * m.value * 1e6 / wave_items, 2) if wave_items > 0 else None}
    return out


def valu_probe_pairs(ctx):
    """gl355_valu_probe_pairs: {(form_x, form_y): shader cycles one X and one Y take together} over the twelve forms of tools/gen_valu_pairs.py"""
    lib = ctx.lib
    n = 0
    names = []
    a, b = C.c_char_p(), C.c_char_p()
    while lib.gl355_valu_probe_pair_names(n, C.byref(a), C.byref(b)) == 0:
        names.append((a.value.decode(), b.value.decode()))
        n += 1
    rates = (C.c_double * n)()
    mhz = (C.c_double * n)()
    ctx.check(lib.gl355_valu_probe_pairs(ctx.h, rates, mhz))
    return {"%s+%s" % names[i]: round(2 * mhz[i] * 1e6 * N_SIMD / (rates[i] * 1e9), 3) for i in range(n) if rates[i] > 0}


# opcode form (gl355_valu_probe_op_name / tools/isa_mix.py `forms`) -> the pair class it overlaps like (tools/gen_valu_pairs.py FORMS)
PAIR_CLASS = {"v_add_u32": "add_u32", "v_sub_u32": "add_u32", "v_and_b32": "and_b32", "v_lshrrev_b32": "ashrrev_i32", "v_ashrrev_i32": "ashrrev_i32",
              "v_mov_b32": "mov_b32", "v_lshlrev_b32": "lshlrev_b32", "v_alignbit_b32": "lshlrev_b32", "v_add3_u32": "lshlrev_b32",
              "v_mul_lo_u32": "lshlrev_b32", "v_add_co_u32 sgpr": "add_co", "v_sub_co_u32 sgpr": "sub_co", "v_addc_co_u32 sgpr": "addc",
              "v_subb_co_u32 sgpr": "subb", "v_cndmask_b32 sgpr": "cndmask_const", "v_cndmask_b32 0,-1,sgpr": "cndmask_const",
              "v_mad_u64_u32 vvv": "mad", "v_mad_u64_u32 svv": "mad", "v_mad_u64_u32 vcv": "mad", "v_mad_u64_u32 v,-1,v": "mad",
              "v_lshl_add_u64": "lshl_add_u64", "v_lshlrev_b64": "lshl_add_u64", "v_lshrrev_b64": "lshl_add_u64", "v_mov_b64": "lshl_add_u64",
              "v_cmp_lt_u64 sgpr": "lshl_add_u64"}
PAIR_FORM_OF_CLASS = {"add_u32": "v_add_u32", "and_b32": "v_and_b32", "ashrrev_i32": "v_ashrrev_i32", "mov_b32": "v_mov_b32", "lshlrev_b32": "v_lshlrev_b32",
                      "add_co": "v_add_co_u32 sgpr", "sub_co": "v_sub_co_u32 sgpr", "addc": "v_addc_co_u32 sgpr", "subb": "v_subb_co_u32 sgpr",
                      "cndmask_const": "v_cndmask_b32 0,-1,sgpr", "mad": "v_mad_u64_u32 svv", "lshl_add_u64": "v_lshl_add_u64"}


def valu_costs(form_counts, ops, pairs=None):
    """SIMD cycles a multiset of VALU instructions needs, two ways.  form_counts: {opcode form: count} (any positive weights).
    additive   sum_f n_f x the form's stand-alone cost (gl355_valu_probe_ops, best of ILP 1 / 4 / 8); forms without a probe at the cheapest measured cost
    floor      the pair-aware lower bound: instructions are matched up (a linear program over the twelve pair classes) so that every matched pair
               (X, Y) costs what the pair probe measured for one X and one Y together, unmatched ones their stand-alone cost; no schedule of this
               multiset can issue faster if overlaps are pairwise -- this prices the ceiling
    -> (additive cycles, floor cycles or None without the pair table, weight of forms without a probe)"""
    iso = {f: v["clk"] for f, v in ops.items() if v.get("clk")}
    cheapest = min(iso.values())
    cls_iso = {c: iso[f] for c, f in PAIR_FORM_OF_CLASS.items() if f in iso}
    n_cls, c_cls, add, unprobed = {}, {}, 0.0, 0.0
    for f, n in form_counts.items():
        if n <= 0:
            continue
        c = iso.get(f)
        cls = PAIR_CLASS.get(f)
        if c is None or cls is None:
            unprobed += n
            c, cls = cheapest, "mov_b32"
        add += n * c
        n_cls[cls] = n_cls.get(cls, 0.0) + n
        c_cls[cls] = c_cls.get(cls, 0.0) + n * c          # an unmatched instruction costs what its OWN form costs alone
    if not pairs:
        return add, None, unprobed
    try:
        from scipy.optimize import linprog
    except Exception:
        return add, None, unprobed
    classes = sorted(n_cls)
    idx = {c: i for i, c in enumerate(classes)}
    var, cost = [], []
    for i, a in enumerate(classes):
        for b in classes[i + 1:]:
            t = pairs.get("%s+%s" % (a, b), pairs.get("%s+%s" % (b, a)))
            if t is not None and a in cls_iso and b in cls_iso and t < c_cls[a] / n_cls[a] + c_cls[b] / n_cls[b]:
                var.append((a, b)); cost.append(t)
    for a in classes:
        var.append((a, None)); cost.append(c_cls[a] / n_cls[a])
    A = [[0.0] * len(var) for _ in classes]
    for k, (a, b) in enumerate(var):
        A[idx[a]][k] = 1.0
        if b is not None:
            A[idx[b]][k] = 1.0
    res = linprog(cost, A_eq=A, b_eq=[n_cls[c] for c in classes], bounds=[(0, None)] * len(var), method="highs")
    return add, (float(res.fun) if res.success else None), unprobed


# how the SQ counters classify the opcode forms (rocprofv3 --pmc over the probe kernels themselves: profiles/r06_valu_probe_pmc.txt)
COUNTER_CLASS = {"int64": ("v_mad_u64_u32 vvv", "v_mad_u64_u32 svv", "v_mad_u64_u32 vcv", "v_mad_u64_u32 v,-1,v", "v_lshl_add_u64", "v_cmp_lt_u64 sgpr"),
                 "int32": ("v_add_u32", "v_sub_u32", "v_ashrrev_i32", "v_add3_u32", "v_mul_lo_u32", "v_add_co_u32 sgpr", "v_sub_co_u32 sgpr",
                           "v_addc_co_u32 sgpr", "v_subb_co_u32 sgpr")}
_CLASS_OF_FORM = {f: c for c, fs in COUNTER_CLASS.items() for f in fs}


class ValuModel:
    """The VALU ceiling of round 6: issue costs per opcode form and per pair of forms MEASURED in this run (gl355_valu_probe_ops /
    _pairs), instruction counts per kernel from the committed steady-state --pmc pass, the split into forms from the shipped ISA (tools/isa_mix.py),
    moved to each kernel's dynamic SQ_INSTS_VALU_INT64 / _INT32 shares.  cycles(forms) -> (additive, pair-aware floor); the floor prices the peak."""

    def __init__(self, ctx, run_pairs=True):
        self.ops = valu_probe_ops(ctx)
        self.pairs = valu_probe_pairs(ctx) if run_pairs else None
        self.composites = valu_probe_composites(ctx)
        self.classes = valu_probe(ctx)
        try:
            self.isa_path, self.pmc_path = latest_profile("_isa_mix.json"), latest_profile("_pmc_traffic.json")
            self.isa = json.load(open(self.isa_path))["kernels"]
            self.pmc = json.load(open(self.pmc_path))
        except Exception:
            self.isa, self.pmc = {}, {}

    def static_forms(self, name):
        k = self.isa.get(name)
        if k is None:                       # template instances of one name: summed
            acc = {}
            for n, v in self.isa.items():
                if n.split("<")[0] == name.split("<")[0]:
                    for f, c in v.get("forms", {}).items():
                        acc[f] = acc.get(f, 0) + c
            return acc or None
        return dict(k.get("forms", {})) or None

    def dynamic_forms(self, name, n=None, n64=None, n32=None):
        """the kernel's static form histogram scaled to `n` dynamic instructions, the INT64- / INT32-counted forms to the counters' dynamic totals"""
        st = self.static_forms(name)
        if not st:
            return None
        tot = float(sum(st.values()))
        if n is None:
            return {f: c / tot for f, c in st.items()}
        if n64 is None or n32 is None:
            return {f: n * c / tot for f, c in st.items()}
        grp = {"int64": {}, "int32": {}, "other": {}}
        for f, c in st.items():
            grp[_CLASS_OF_FORM.get(f, "other")][f] = c
        want = {"int64": n64, "int32": n32, "other": max(0.0, n - n64 - n32)}
        out = {}
        for g, forms in grp.items():
            t = float(sum(forms.values()))
            if t <= 0:
                if want[g] > 0:             # the counter saw instructions of a class the static body does not have: the class's commonest form
                    f = {"int64": "v_mad_u64_u32 vvv", "int32": "v_add_u32", "other": "v_mov_b32"}[g]
                    out[f] = out.get(f, 0.0) + want[g]
                continue
            for f, c in forms.items():
                out[f] = out.get(f, 0.0) + want[g] * c / t
        return out

    def job_forms_per_unit(self):
        """{form: dynamic instructions per unit} over every kernel of the steady-state pass, and the instruction total"""
        job = self.pmc.get("job", {})
        n_units = job.get("units_steady")
        if not n_units:
            return None, None
        acc, total = {}, 0.0
        for name, e in self.pmc["kernels"].items():
            n = e.get("steady_valu_insts")
            if not n or n <= 0 or name.startswith("vp"):
                continue
            d = self.dynamic_forms(name, n / n_units, e.get("steady_valu_int64", 0.0) / n_units, e.get("steady_valu_int32", 0.0) / n_units)
            if d is None:
                d = {"unprobed:%s" % name: n / n_units}
            for f, c in d.items():
                acc[f] = acc.get(f, 0.0) + c
            total += n / n_units
        return acc, total

    def cycles(self, forms):
        return valu_costs(forms, self.ops, self.pairs)

    def peak(self, forms, clock_mhz):
        """-> {additive / floor: G wave-instructions per second the chip could issue for this multiset at `clock_mhz`, clk per instruction}"""
        n = float(sum(forms.values()))
        add, fl, unp = self.cycles(forms)
        r = {"clk_per_inst_additive": round(add / n, 3), "peak_additive": round(N_SIMD * clock_mhz * 1e6 / (add / n) / 1e9, 1), "unprobed_share": round(unp / n, 4)}
        if fl:
            r.update({"clk_per_inst_floor": round(fl / n, 3), "peak": round(N_SIMD * clock_mhz * 1e6 / (fl / n) / 1e9, 1)})
        else:
            r["peak"] = r["peak_additive"]
        return r

    def composite_checks(self):
        """the shipped product and permutation on registers against their own floors: achieved <= ceiling must hold (it is how the model is audited in
        the run that uses it); instruction counts per item from the committed --pmc pass over the probe kernels"""
        per_item = self.pmc.get("probes", {})
        out = {}
        for comp, kern in (("product_x4_lockstep", "vpc_product_kernel"), ("poseidon_permutation", "vpc_permute_kernel")):
            c = self.composites.get(comp)
            st = self.static_forms(kern)
            n = per_item.get(kern, {}).get("valu_insts_per_item")
            if not (c and st and n):
                continue
            e = per_item[kern]
            forms = self.dynamic_forms(kern, n, e.get("valu_int64_per_item"), e.get("valu_int32_per_item"))
            add, fl, _ = self.cycles(forms)
            meas = c["simd_clk_per_wave_item"]
            out[comp] = {"measured_simd_clk_per_wave_item": meas, "valu_insts_per_item": n, "additive_model_clk": round(add, 1),
                         "floor_clk": round(fl, 1) if fl else None, "waves_per_simd": c["waves_per_simd"],
                         "measured_over_floor": round(meas / fl, 4) if fl else None, "ceiling_holds": bool(fl is None or meas >= fl)}
        return out

    def report(self):
        iso = {f: v["clk"] for f, v in self.ops.items()}
        over = {}
        if self.pairs:
            ci = {c: iso.get(f) for c, f in PAIR_FORM_OF_CLASS.items()}
            for k, t in self.pairs.items():
                a, b = k.split("+")
                if ci.get(a) and ci.get(b) and t < 0.97 * (ci[a] + ci[b]):
                    over[k] = {"together": t, "alone": round(ci[a] + ci[b], 3)}
        return {"clk_per_wave_inst_per_simd_by_opcode_form": self.ops, "pairs_that_overlap": over,
                "pairs_measured": len(self.pairs) if self.pairs else 0, "composites": self.composites, "composite_checks": self.composite_checks(),
                "classes_round5": self.classes,
                "sources": [os.path.basename(x) for x in (self.isa_path, self.pmc_path) if x]}


VALU_MODEL = None


class ClockSampler:
    """shader clock during the timed region: gl355_clock_probe (one sleeping wave for 2 ms) on a context of its own every ~100 ms"""

    def __init__(self, gl, device, period=0.1):
        import threading
        self.ctx = gl.Context(device)
        self.period = period
        self.samples, self.stop = [], threading.Event()
        self.thread = threading.Thread(target=self.run, daemon=True)

    def run(self):
        v = C.c_double(0)
        while not self.stop.is_set():
            if self.ctx.lib.gl355_clock_probe(self.ctx.h, 2000, C.byref(v)) == 0 and v.value > 0:
                self.samples.append(v.value)
            self.stop.wait(self.period)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.thread.join()
        self.ctx.close()

    def summary(self):
        if not self.samples:
            return None
        # a probe wave descheduled in mid-sleep (two processes on one device) reads a nonsense ratio: samples beyond 1.25x the
        # median are dropped and counted
        med = sorted(self.samples)[len(self.samples) // 2]
        kept = [s for s in self.samples if s <= 1.25 * med]
        return {"mean_mhz": round(sum(kept) / len(kept)), "min_mhz": round(min(kept)), "max_mhz": round(max(kept)),
                "samples": len(kept), "dropped": len(self.samples) - len(kept)}


def valu_mix(kernel=None):
    """Instruction-class fractions of `kernel` (None: of the whole unit, every kernel weighted by its dynamic instruction count).
    mad64 share: dynamic, SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU of the committed --pmc pass when it carries those counters; the rest
    splits into full32 / half32 as the kernel's shipped ISA does (profiles/rNN_isa_mix.json, tools/isa_mix.py).  Without the INT64
    counters everything comes from the static histogram.  -> (fractions, dynamic VALU instructions per launch or per unit, source)"""
    try:
        pmc = json.load(open(latest_profile("_pmc_traffic.json")))
        isa = json.load(open(latest_profile("_isa_mix.json")))["kernels"]
    except Exception:
        return None, None, None
    src = "profiles/%s + profiles/%s" % (os.path.basename(latest_profile("_pmc_traffic.json")), os.path.basename(latest_profile("_isa_mix.json")))

    def static_f(name):
        k = isa.get(name)
        if k is None:      # template instances: name<...>
            cands = [v for n, v in isa.items() if n.split("<")[0] == name]
            if not cands:
                return None
            tot = sum(v["valu_static"] for v in cands)
            return {c: sum(v[c] for v in cands) / tot for c in VALU_CLASSES}
        return dict(k["f"])

    def one(name, e):
        f = static_f(name.split("<")[0]) or {"full32": 0.11, "half32": 0.36, "mad64": 0.53}
        n = e.get("valu_insts_per_launch")
        i64 = e.get("valu_int64_per_launch")
        if n and i64 is not None:
            rest = f["full32"] + f["half32"]
            m = i64 / n
            f = {"mad64": m, "full32": (1 - m) * f["full32"] / rest, "half32": (1 - m) * f["half32"] / rest}
        return f, n
    if kernel is not None:
        e = pmc["kernels"].get(kernel)
        if not e or not e.get("valu_insts_per_launch"):
            return None, None, None
        f, n = one(kernel, e)
        return {c: round(f[c], 4) for c in VALU_CLASSES}, n, src
    job = pmc.get("job")
    if not job:
        return None, None, None
    tot, acc = 0.0, {c: 0.0 for c in VALU_CLASSES}
    for name, e in pmc["kernels"].items():
        if not e.get("valu_insts_per_launch") or name.startswith("vp_"):
            continue
        f, n = one(name, e)
        w = n * e.get("sq_launches", e.get("launches", 0))
        tot += w
        for c in VALU_CLASSES:
            acc[c] += w * f[c]
    if tot <= 0:
        return None, None, None
    return {c: round(acc[c] / tot, 4) for c in VALU_CLASSES}, job["valu_insts_per_unit"], src


NOMINAL_CLK = {"full32": 2.0, "half32": 4.0, "mad64": 4.0}


def valu_peak(mix, clock_mhz):
    """G wave-instructions/s the chip can issue at `clock_mhz` for instructions that split as `mix`: 1024 SIMDs x clock / the mix-weighted
    issue cost.  No kernel can exceed it (every class priced at the hardware's issue rate), so achieved / peak <= 1 by construction."""
    cost = sum(mix[c] * NOMINAL_CLK[c] for c in VALU_CLASSES)
    return N_SIMD * clock_mhz * 1e6 / cost / 1e9


def valu_peak_probe(mix, classes, clock_mhz=None):
    """the same with the class rates the probe kernels reached in this run (moved to `clock_mhz` if given)"""
    t = 0.0
    for c in VALU_CLASSES:
        r = classes[c]["rate_ginst_s"]
        if clock_mhz and classes[c]["shader_mhz"]:
            r = r * clock_mhz / classes[c]["shader_mhz"]
        if r <= 0:
            return None
        t += mix[c] / r
    return 1.0 / t if t > 0 else None


class _TorchComm:
    """stand-in with the Comm interface over torch.distributed -- used ONLY if some rank cannot bind librccl for the gl355
    communicator (reported in the JSON line as config.exchange); the product's exchange is gl355_gather_digests"""

    def __init__(self, dist, dev):
        self.dist, self.dev, self.backend_name = dist, dev, "torch.distributed (gl355 RCCL communicator unavailable on some rank)"

    def gather(self, local):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint64).view(np.int64)).to(self.dev)
        parts = [torch.empty_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, t)
        return torch.cat(parts, dim=0).cpu().numpy().view(np.uint64)

    def barrier(self):
        self.dist.barrier()

    def max(self, v):
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.dist.destroy_process_group()


def thread_cpu_snapshot():
    """{tid: (comm, cpu seconds)} of the process's live threads (diagnostic: which threads burn host CPU; GL355_BENCH_THREAD_CPU=1)"""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tck)
        except Exception:
            pass
    return out


_STORE_KEEPALIVE = []


def open_comm(lib, par, ctx, rank, world, rehearsal, dev=None):
    """The exchange of the N > 1 job through the C ABI (gl355_comm_*): RCCL over xGMI, or TCP between the host processes in the
    one-device rehearsal.  The 128-byte communicator id travels through the launcher's key-value store (torchrun's TCPStore,
    MASTER_ADDR / MASTER_PORT) -- plumbing a Rust host would do with its own launcher; no torch collective is involved."""
    if world == 1:
        return None
    from datetime import timedelta
    from torch.distributed import PrefixStore, TCPStore
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = TCPStore(addr, port, world, (rank == 0 and not agent_store), timedelta(seconds=300), multi_tenant=True)
    # rank 0 may be the store's server (no agent store): the server must outlive every other rank's reads of the flags below, so the object is
    # kept for the life of the process (all ranks pass a communicator barrier before they exit)
    _STORE_KEEPALIVE.append(store)
    store = PrefixStore("gl355_bench/%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), store)
    backend = par.COMM_HOST if rehearsal else par.COMM_RCCL
    # phase 1: can every rank bind its communicator library?  (a rank that cannot must not leave the others inside ncclCommInitRank)
    ok, cid, why = True, b"", ""
    try:
        if backend == par.COMM_HOST:
            import socket
            if rank == 0:
                s = socket.socket(); s.bind((addr if addr[0].isdigit() else "127.0.0.1", 0)); hp = s.getsockname()[1]; s.close()
                cid = par.Comm.unique_id(lib, backend, addr if addr[0].isdigit() else "127.0.0.1", hp)
        else:
            cid = par.Comm.unique_id(lib, backend)          # every rank: proves librccl binds here; rank 0's id is the one used
    except Exception as exc:
        ok, why = False, repr(exc)
    store.set("ok/%d" % rank, b"1" if ok else why.encode()[:200] or b"0")
    if rank == 0 and ok:
        store.set("id", cid)
    flags = [bytes(store.get("ok/%d" % r)) for r in range(world)]
    created_failed = False
    if all(f == b"1" for f in flags):
        # phase 2: the communicator itself (ncclCommInitRank is collective).  Every rank reports whether it came up; the job uses it only if
        # all did -- a communicator that exists on some ranks only would hang the first gather
        comm, err = None, ""
        try:
            if os.environ.get("GL355_BENCH_FORCE_COMM_FAIL") == "1":      # test hook for the fall-back below
                raise RuntimeError("forced failure (GL355_BENCH_FORCE_COMM_FAIL)")
            comm = par.Comm(ctx, backend, bytes(store.get("id")), rank, world, lib=lib)
        except Exception as exc:
            err = repr(exc)
        store.set("up/%d" % rank, b"1" if comm is not None else (err.encode()[:200] or b"0"))
        flags = [bytes(store.get("up/%d" % r)) for r in range(world)]
        if all(f == b"1" for f in flags):
            comm.backend_name = "gl355_gather_digests over RCCL (ncclAllGather)" if backend == par.COMM_RCCL else "gl355_gather_digests over TCP (one-device rehearsal)"
            return comm
        if comm is not None:
            comm.close()
        created_failed = True
    why = [f for f in flags if f != b"1"][:1]
    # --gpus N > 1 measures the RCCL exchange of SURVEY 8(e): when the RCCL communicator cannot be created on every rank the run FAILS
    # (every rank sees the same flags, so every rank exits) instead of quietly measuring something else.  GL355_BENCH_ALLOW_STANDIN=1
    # (never set by the driver) lets the 64-byte-per-unit exchange run over torch.distributed instead; the line then says so in
    # config.exchange.  The one-device rehearsal never had RCCL to begin with.
    if not rehearsal and os.environ.get("GL355_BENCH_ALLOW_STANDIN") != "1":
        sys.stderr.write("[bench] rank %d: the RCCL communicator (gl355_comm_create) is unavailable: %s -- refusing to substitute another "
                         "exchange (set GL355_BENCH_ALLOW_STANDIN=1 to allow the torch.distributed stand-in)\n" % (rank, why))
        sys.stderr.flush()
        os._exit(3)
    sys.stderr.write("[bench] rank %d: gl355 communicator unavailable (%s); using torch.distributed for the exchange\n" % (rank, why))
    import torch.distributed as dist
    # the exchange is 64 bytes per unit: when the RCCL communicator could not be CREATED (rather than librccl not binding), RCCL itself is
    # suspect, so the stand-in runs over gloo on host tensors
    if rehearsal or created_failed:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        c = _TorchComm(dist, "cpu")
        c.backend_name = "torch.distributed gloo on host tensors (gl355 communicator could not be created on some rank)" if created_failed else c.backend_name
        return c
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return _TorchComm(dist, dev)
