"""The VALU cost model of bench.py (tools/bench_common.py ValuModel / valu_costs) on the committed probe record (no GPU): the pair-aware floor is a lower
bound of the additive sum, pairing never prices an instruction below what a probe measured, the job's instruction multiset adds up to the steady-state
count of the committed --pmc pass, and the two composite probes of that record do not beat their own floors (the audit bench.py repeats in every run)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_common as bc  # noqa: E402


@pytest.fixture(scope="module")
def model():
    path = bc.latest_profile("_valu_probe.json")
    if not path:
        pytest.skip("no committed probe record")
    d = json.load(open(path))
    m = bc.ValuModel.__new__(bc.ValuModel)
    m.ops, m.pairs, m.composites, m.classes = d["ops"], d.get("pairs"), d["composites"], d["classes"]
    m.isa_path, m.pmc_path = bc.latest_profile("_isa_mix.json"), bc.latest_profile("_pmc_traffic.json")
    m.isa = json.load(open(m.isa_path))["kernels"]
    m.pmc = json.load(open(m.pmc_path))
    return m


def test_floor_is_below_the_additive_sum_and_above_the_cheapest_form(model):
    for kern in ("hash_leaves_kernel", "quotient_kernel<2, false>", "ntt_rows_l24s_kernel<5, false>", "merkle_level_kernel"):
        forms = model.dynamic_forms(kern)
        add, fl, unprobed = model.cycles(forms)
        assert fl is not None and 0 < fl <= add * (1 + 1e-9), kern
        cheapest = min(v["clk"] for v in model.ops.values())
        assert fl >= 0.5 * cheapest * sum(forms.values()), kern          # a pair never costs less than one instruction of the cheapest form
        assert unprobed / sum(forms.values()) < 0.05, kern


def test_pairing_uses_only_measured_overlaps(model):
    # two multiply-adds never overlap: a multiset of nothing but multiply-adds is priced additively
    forms = {"v_mad_u64_u32 vvv": 1000.0}
    add, fl, _ = model.cycles(forms)
    assert abs(add - fl) < 1e-6 * add
    # a select on constants next to a carry step is cheaper together than apart (the overlap the lock-step product lives on)
    forms = {"v_cndmask_b32 0,-1,sgpr": 500.0, "v_sub_co_u32 sgpr": 500.0}
    add, fl, _ = model.cycles(forms)
    assert fl < 0.75 * add


def test_job_multiset_matches_the_steady_state_count(model):
    forms, total = model.job_forms_per_unit()
    assert forms and abs(sum(forms.values()) - total) < 1e-6 * total
    assert abs(total - model.pmc["job"]["valu_insts_per_unit"]) < 2
    assert model.pmc["job"]["valu_insts_per_unit"] < model.pmc["job"]["valu_insts_per_unit_including_setup"]
    pk = model.peak(forms, 2300.0)
    assert pk["peak"] >= pk["peak_additive"] and 2.0 < pk["clk_per_inst_floor"] < pk["clk_per_inst_additive"] < 4.3


def test_composite_probes_do_not_beat_their_floors(model):
    checks = model.composite_checks()
    assert set(checks) == {"product_x4_lockstep", "poseidon_permutation"}
    for name, c in checks.items():
        assert c["ceiling_holds"] and c["measured_over_floor"] >= 1.0, (name, c)
        assert c["measured_simd_clk_per_wave_item"] <= 1.1 * c["additive_model_clk"], (name, c)
