"""gl355_valu_probe / gl355_clock_probe on cuda:0, printed: the class rates the VALU roofline of bench.py uses.  Under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64` the same run calibrates how the counters classify the three probe
instructions (tools/prof_round4.sh -> profiles/r04_valu_probe_pmc.txt)."""
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

gl = importlib.import_module("stark-verifier_amd")
ctx = gl.Context(0)
out = {"classes": bench.valu_probe(ctx)}
v = C.c_double(0)
ctx.check(ctx.lib.gl355_clock_probe(ctx.h, 2000, C.byref(v)))
out["idle_clock_mhz"] = round(v.value)
print(json.dumps(out))
ctx.close()
