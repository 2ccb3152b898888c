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
