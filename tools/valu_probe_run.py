"""gl355_valu_probe / gl355_valu_probe_ops / gl355_valu_probe_composite / gl355_clock_probe on cuda:0, printed as one JSON document: the issue
costs the VALU roofline of bench.py is priced with.  Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64` the same
run gives the dynamic instruction counts of the composite probes (instructions per product / per permutation) and shows how the counters classify
each probe instruction (tools/prof_round6.sh -> profiles/r06_valu_probe_pmc.txt)."""
import ctypes as C
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_common as bc  # noqa: E402

gl = importlib.import_module("stark-verifier_amd")
ctx = gl.Context(0)
out = {"classes": bc.valu_probe(ctx)}
out["ops"] = bc.valu_probe_ops(ctx)
out["composites"] = bc.valu_probe_composites(ctx)
out["pairs"] = bc.valu_probe_pairs(ctx)
v = C.c_double(0)
ctx.check(ctx.lib.gl355_clock_probe(ctx.h, 2000, C.byref(v)))
out["idle_clock_mhz"] = round(v.value)
print(json.dumps(out, indent=1))
ctx.close()
