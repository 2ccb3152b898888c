"""Does a device wait sleep or spin?  wall vs process CPU time around 5 x 2^24 Poseidon permutations (~38 ms), for
hipStreamSynchronize / blocking-event waits, with and without hipSetDeviceFlags(hipDeviceScheduleBlockingSync) before the
HIP runtime is initialised.  usage: python tools/blocking_wait_probe.py [devflag]"""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if len(sys.argv) > 1:
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    print("hipSetDeviceFlags(%s) ->" % sys.argv[1], hip.hipSetDeviceFlags(int(sys.argv[1], 0)))
gl = importlib.import_module("stark-verifier_amd")
for blocking in (0, 1):
    ctx = gl.Context(0)
    if blocking:
        ctx.set_option(2, 1)
    x = torch.randint(0, 1 << 62, (1 << 24, 12), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):
        w0, c0 = time.perf_counter(), time.process_time()
        for _ in range(5):
            ctx.check(ctx.lib.gl355_poseidon_permute(ctx.h, C.c_void_p(x.data_ptr()), 1 << 24))
        ctx.sync()
        w1, c1 = time.perf_counter(), time.process_time()
    print("GL355_OPT_BLOCKING_SYNC=%d: wall %.1f ms, cpu %.1f ms" % (blocking, 1e3 * (w1 - w0), 1e3 * (c1 - c0)))
