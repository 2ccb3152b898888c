"""Build-time check (no GPU): in the gfx950 ISA of the kernels with hand-placed carries, every VALU read of a VALU-written scalar register keeps the
two wait states the hardware needs (tools/check_hazards.py; ADVICE r5 #1 -- the lock-step products rely on the ORDER of separate inline-asm statements,
which LLVM's hazard recogniser cannot see into).  The negative control compiles the same source with the wait states compiled out and must be flagged."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_hazards as ch  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles gfx950 without a GPU)")


@pytest.mark.parametrize("src", ["merkle.hip", "quotient.hip", "fri.hip"])
def test_valu_scalar_hazard_distance(src):
    bad, n = ch.check_listing(ch.listing_of(src), src)
    assert n > 500, "the checker saw almost no carry chains in %s: the ISA parser is out of date" % src
    assert not bad, "VALU reads of a VALU-written scalar with < 2 wait states: %s" % (bad[:5],)


def test_checker_flags_a_build_without_wait_states():
    bad, n = ch.check_listing(ch.listing_of("merkle.hip", ("-DGL_EXPERIMENT_NO_HAZARD_NOP",)), "merkle.hip")
    assert len(bad) > 100
