"""GPU: parity under load, collected at reduced size (tests/parity_sweep.py is the hand-run full sweep): 64 random units of the
depth-20 workload on 22 concurrent prover contexts == the same units one by one on one context (determinism under concurrency,
the reference proves from rayon workers: recursion.rs:214-227,300-308), and 4 sampled units byte-identical -- Semaphore proof and
recursive proof -- to the CPU restatement of prove()."""
import pytest

pytestmark = pytest.mark.gpu


def test_parity_sweep_64_units_22_contexts():
    import parity_sweep
    lines = []
    parity_sweep.run_sweep(M=64, K=4, contexts=22, blocking_sync=True, out=lines.append)
    assert any(l.startswith("PARITY SWEEP OK") for l in lines), "\n".join(lines)


def test_cfg5_per_gpu_share_128_units_8_contexts():
    """BASELINE configs[4]: 1024 proofs over 8 GPUs = 128 units per GPU.  One GPU's whole step exactly as bench.py runs it (8 prover contexts
    x 8 lock-step units through gl355_semaphore_units): deterministic against the one-context run, nullifiers == hash(sk | topic) and the
    aggregation root == the oracle's Merkle root over the 128 leaves, every recursive proof accepted by gl355_verify, one unit byte-identical
    to the CPU prover (the largest batch any test proved was 64 units)"""
    import parity_sweep
    lines = []
    parity_sweep.run_sweep(M=128, K=1, contexts=8, blocking_sync=2, out=lines.append, check_root=True)
    assert any(l.startswith("PARITY SWEEP OK") for l in lines), "\n".join(lines)
    assert any("aggregation root over 128 leaves == oracle" in l for l in lines), "\n".join(lines)
