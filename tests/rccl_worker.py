"""One rank of the world > 1 RCCL test (tests/test_gpu_comm.py::test_rccl_two_devices): device = rank, gl355_comm_create(RCCL) with the
id handed over through a file, gl355_gather_digests of this rank's block of (nullifier | topic)-shaped leaves with host and with device
operands, barrier, max, and on rank 0 the aggregation root of everything gathered (printed)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def block(rank, per):
    rng = np.random.default_rng(0xC0DE + rank)
    return rng.integers(0, 0xFFFFFFFF00000001, size=(per, 8), dtype=np.uint64)


def main():
    rank, world, idfile, per = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    import torch
    torch.cuda.set_device(rank)
    gl = importlib.import_module("stark-verifier_amd")
    par = importlib.import_module("stark-verifier_amd.parallel")
    ctx = gl.Context(rank)
    if rank == 0:
        cid = par.Comm.unique_id(ctx.lib, par.COMM_RCCL)
        with open(idfile + ".tmp", "wb") as f:
            f.write(cid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no communicator id after 120 s" % rank)
            time.sleep(0.05)
        cid = open(idfile, "rb").read()
    comm = par.Comm(ctx, par.COMM_RCCL, cid, rank, world)
    mine = block(rank, per)
    allv = comm.gather(mine)                                                        # host operands, staged through HBM
    want = np.concatenate([block(r, per) for r in range(world)])
    assert np.array_equal(allv, want), "rank %d: host-operand gather differs" % rank
    t = torch.from_numpy(mine.view(np.int64)).cuda()
    out = torch.zeros((world * per, 8), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert ctx.lib.gl355_gather_digests(comm.h, t.data_ptr(), mine.size, out.data_ptr()) == 0     # device operands
    ctx.sync()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want), "rank %d: device-operand gather differs" % rank
    comm.barrier()
    assert comm.max(1.5 + rank) == 1.5 + world - 1
    if rank == 0:
        root = par.aggregation_root(ctx, allv)
        print("ROOT " + " ".join("%016x" % int(v) for v in root.reshape(-1)), flush=True)
    comm.barrier()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
