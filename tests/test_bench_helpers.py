"""bench.py's host-side helpers (no GPU): the priced-issue arithmetic of the roofline, the core count and where it came from,
the GPU count read without the HIP runtime, and the shape of the multi-rank spread."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_priced_issue_prices_every_class_and_brackets_the_rest():
    b = _bench()
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_c2.json")))
    c = pmc["counters"]
    simd_cycles = 1024 * 2.4e9 * 13.6e-6
    pr = b.priced_issue(c, simd_cycles)
    classed = sum(c[k] for k in b.VALU_ISSUE_CYCLES)
    other = c["SQ_INSTS_VALU"] - classed
    want = sum(c[k] * v for k, v in b.VALU_ISSUE_CYCLES.items()) + 3 * other
    assert abs(pr["valu_issue_cycles"] - want) < 1e-6
    assert pr["frac_other_at_2"] < pr["frac"] < pr["frac_other_at_4"]
    assert 0.2 < pr["frac"] < 0.4 and 0.05 < pr["fp64_pipe_frac"] < pr["frac"]  # (C2: about 0.3 and 0.13)
    assert b.priced_issue({"SQ_INSTS_VALU": 10.0}, simd_cycles) is None  # no class counters: not priced
    # HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE in KiB (gfx950 tallies 128-B reads at 64 B)
    assert pmc["hbm_bytes_per_launch"] == round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)


def test_core_and_gpu_counts_say_where_they_come_from():
    b = _bench()
    n, source = b.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and isinstance(source, str) and source
    g = b.visible_gpus()
    assert g is None or (isinstance(g, int) and g >= 1)  # None: no KFD topology here (no GPU in the build container)
    assert os.environ.get("HIP_FORCE_DEV_KERNARG") == "1" or "HIP_FORCE_DEV_KERNARG" in os.environ  # set before any HIP call


def test_profiles_are_priced_only_on_the_binary_they_measured(tmp_path, monkeypatch):
    """every PMC summary carries the sha256 of the device code it was collected on
    (pytracer_amd.build.code_hash: the library's .hip_fatbin section), bench.py compares it with the library it loaded and
    prices nothing from a file of another build."""
    from pytracer_amd import _lib, build

    b = _bench()
    h = build.code_hash(_lib.lib_path())
    assert len(h) == 64 and h == build.code_hash(_lib.lib_path()) and b.loaded_code_hash() == h
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    (prof / "same.json").write_text(json.dumps({"code_hash": h, "counters": {"SQ_INSTS_VALU": 1.0}}))
    (prof / "other.json").write_text(json.dumps({"code_hash": "0" * 64, "counters": {"SQ_INSTS_VALU": 1.0}}))
    (prof / "round4.json").write_text(json.dumps({"counters": {"SQ_INSTS_VALU": 1.0}}))  # no hash at all
    assert b.fresh(b.load_profile("same.json")) and b.load_profile("same.json")["counters"]["SQ_INSTS_VALU"] == 1.0
    for stale in ("other.json", "round4.json"):
        got = b.load_profile(stale)
        assert not b.fresh(got) and "not priced" in got["stale"] and "counters" not in got
    assert b.load_profile("missing.json") is None and not b.fresh(None)
    assert b.load_profile("other.json", want_hash=None)["counters"]  # (read without the check: tools only)
    with __import__("pytest").raises(ValueError):
        build.code_hash(__file__)  # not an ELF file


def test_counter_cross_checks_and_scalar_co_issue():
    b = _bench()
    c = {k: 0.0 for k in b.VALU_ISSUE_CYCLES}
    c.update(SQ_INSTS_VALU=1000.0, SQ_INSTS_VALU_ADD_F64=400.0, SQ_INSTS_VALU_ADD_F32=200.0, SQ_INSTS_SALU=500.0,
             SQ_ACTIVE_INST_VALU=1000.0, SQ_WAVE_CYCLES=40000.0, SQ_WAIT_ANY=20000.0, SQ_WAIT_INST_ANY=5000.0)
    pr = b.priced_issue(c, 10000.0)
    assert abs(pr["frac"] - (400 * 4 + 200 * 2 + 400 * 3) / 10000.0) < 1e-12 and abs(pr["unclassed_share"] - 0.4) < 1e-12
    assert abs(pr["frac_counter_active_inst_valu_x4"] - 0.4) < 1e-12  # every instruction charged a quad-cycle
    assert abs(pr["frac_with_salu_coissue"] - (pr["frac"] + 500 * b.SALU_COISSUE_PENALTY_CYCLES / 10000.0)) < 1e-12
    assert pr["wave_cycles_waiting_any_frac"] == 0.5 and pr["wave_cycles_waiting_inst_frac"] == 0.125
