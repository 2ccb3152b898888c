"""CPU: `python bench.py --gpus N` with NO launcher around it starts its N ranks itself (the driver runs the N = 1 line as plain
`python3 bench.py --gpus 1`; the N = 8 line must be launchable the same way).  The ranks rendezvous on 127.0.0.1, exchange through the C ABI's
communicator and rank 0 alone prints the JSON line; a rank that dies ends the others and the launcher's exit code is non-zero."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(180)
def test_self_launch_exchange_four_ranks():
    r = _run(["--gpus", "4", "--workload", "exchange", "--steps", "3", "--warmup", "1", "--proofs-per-step", "128"], {"GL355_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the JSON line: %r" % r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["leaves_in_rank_order"] is True and line["launcher"] == "bench.py itself"
    assert "TCP" in line["exchange"]


@pytest.mark.timeout(180)
def test_self_launch_propagates_a_dead_rank():
    # the default workload needs a GPU: on this box every rank fails, the launcher must come back non-zero instead of hanging;
    # on a GPU box with one device rank 1 fails (no cuda:1, and RCCL is never substituted) and takes rank 0 with it
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"GL355_BENCH_ONE_DEVICE": "0"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
