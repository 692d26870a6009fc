"""The LDS bank-conflict model of the line transforms' exchanges (tools/lds/bank_model.py; MI355X_MICROARCH.md, LDS: lane groups and banks
per DS instruction) on the layouts the kernels ship (ocean_fft_core.h, "LDS layout of the exchanges"): no conflict cycle in any step kernel from
256^2 up -- what profiles/r05_lds_conflicts.txt measured with SQ_LDS_BANK_CONFLICT on the MI355X -- and round 4's paddings, on the same model,
at the ratios round 4's counters showed.  Guards the layout formulas against an edit that brings a conflict back; no GPU."""

import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    spec = importlib.util.spec_from_file_location("bank_model", os.path.join(ROOT, "tools", "lds", "bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.QUIET = True
    return m


def test_shipped_layouts_have_no_bank_conflicts(model):
    got = model.new_scheme([256, 512, 1024, 2048, 4096])
    assert len(got) == 10
    for key, (ideal, actual) in got.items():
        assert ideal > 0 and actual == ideal, (key, ideal, actual)


def test_pad0_matches_the_header(model):
    # lds_pad0 of ocean_fft_core.h, restated in the model: the two must not drift apart
    text = open(os.path.join(ROOT, "datum_amd", "csrc", "ocean_fft_core.h")).read()
    assert "constexpr int lds_pad0(int e, int w) { return ((32 / e) / w) > 1 ? (32 / e) / w : 1; }" in text
    for e in (4, 8, 16):
        for w in (1, 2, 4, 8):
            assert model.pad0(e, w) == max(1, (32 // e) // w)


def test_round4_paddings_conflict_as_measured(model):
    # round 4's i + (i >> 4) / i + (i >> 3) paddings on the model: the ratios its counters showed (profiles/r05_lds_conflicts.txt: column pass
    # 4096^2 48.1 %, row pass 1024^2 33.4 % measured) -- the model is what the new layouts were designed on, so it has to reproduce the old ones
    ideal, actual = model.colpass(4096, 8, 3, 2, 12, 2)
    assert abs((actual - ideal) / actual - 0.50) < 0.02
    ideal, actual = model.rowpass(1024, 8, 4, 2)
    assert abs((actual - ideal) / actual - 0.344) < 0.02
