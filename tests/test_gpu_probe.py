"""The measurement entry points of the C ABI (include/gl355.h: gl355_valu_probe_ops / _pairs / _composite, gl355_clock_probe) on the device:
every opcode form reports a plausible issue cost, the pair table covers all 66 pairs, the cost model built from them (tools/bench_common.py ValuModel,
what bench.py prices its roofline with) keeps the run's own composite probes under its ceiling."""
import ctypes as C
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_opcode_and_pair_probes_and_the_ceiling_they_price(gl, ctx):
    import bench_common as bc
    m = bc.ValuModel(ctx)
    assert len(m.ops) == 25 and len(m.pairs) == 66
    for form, v in m.ops.items():
        assert 1.8 < v["clk"] < 5.0, (form, v)                       # 2 clk is the SIMD-32's floor for a wave64 instruction
        assert v["clk"] <= min(v["ilp1"], v["ilp4"], v["ilp8"]) + 1e-9
    full = [m.ops[f]["clk"] for f in ("v_add_u32", "v_and_b32", "v_mov_b32")]
    half = [m.ops[f]["clk"] for f in ("v_mad_u64_u32 vvv", "v_add_co_u32 sgpr", "v_cndmask_b32 0,-1,sgpr", "v_lshlrev_b32")]
    assert max(full) < 2.8 and min(half) > 3.6                       # the two rate classes of gfx950's integer VALU
    # two multiply-adds do not overlap; a select on constants and a carry step do
    iso = {c: m.ops[f]["clk"] for c, f in bc.PAIR_FORM_OF_CLASS.items()}
    assert m.pairs["mad+lshl_add_u64"] > 0.93 * (iso["mad"] + iso["lshl_add_u64"])
    assert m.pairs["cndmask_const+sub_co"] < 0.75 * (iso["cndmask_const"] + iso["sub_co"])
    checks = m.composite_checks()
    assert checks and all(c["ceiling_holds"] for c in checks.values()), checks
    forms, total = m.job_forms_per_unit()
    pk = m.peak(forms, 2300.0)
    assert pk["peak"] >= pk["peak_additive"]
    mhz = C.c_double(0)
    ctx.check(ctx.lib.gl355_clock_probe(ctx.h, 1000, C.byref(mhz)))
    assert 500 < mhz.value < 3000
    # bad arguments are refused, not crashed on
    r = (C.c_double * 25)()
    assert ctx.lib.gl355_valu_probe_ops(ctx.h, 3, r, r) != 0
    assert ctx.lib.gl355_valu_probe_composite(ctx.h, 99, C.byref(mhz), C.byref(mhz), None) != 0
    assert ctx.lib.gl355_valu_probe_op_name(25) is None and ctx.lib.gl355_valu_probe_composite_name(13) is None
