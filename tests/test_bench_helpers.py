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


# ---- the result line: compact, bounded, a fixed set of keys; the rest goes to bench_detail.json ----
N1_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
           "config", "parity_check", "roofline", "cpu_baseline", "ms_per_frame_at_c_abi", "frame_with_dome_shortcut", "detail_file"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "frac_bounds", "traffic", "algorithmic_bytes", "kernel", "avg_kernel_ms",
                 "code_hash_matches_loaded_library"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample", "ms_per_frame", "one_core_Mray_s", "cpu_model"}
LONG = "a note that goes on and on " * 400  # 10 KB: what the old one-line output carried in a dozen places


def _canned_n1():
    return {
        "metric": "Mray/s (primary+shadow; rays handed to a world query) at 1280x720 per GPU, C2", "value": 62103.0491, "unit": "Mray/s",
        "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.014839851, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "code_hash": "ef" * 32, "rays_per_step": 921600, "ray_shape_tests_per_s": 2.05e12,
        "config": {"workload": "C2 flat 1280x720", "width": 1280, "height": 720, "n_shapes": 33, "renderer": "FlatRenderer", "more": LONG},
        "repeats": {"note": LONG}, "value_note": LONG,
        "parity_check": {"bit_identical": True, "pixels": 921600, "against": LONG},
        "frame_with_dome_shortcut": {"ms_per_frame": 0.0132, "rays_traced_per_frame": 509440, "traced_Mray_s": 38600.0, "bit_identical": True, "note": LONG},
        "roofline": {"bound": "valu_issue", "achieved": 14.28, "peak": 42.6, "unit": "T lane-op/s", "frac": 0.335, "traffic": 11664896,
                     "frac_bounds_from_disassembly": [0.3307, 0.3396], "kernel": "pt_tile4_kernel<FLAT> " + LONG, "avg_kernel_ms": 0.01313,
                     "note": LONG, "executed": {"code_hash_matches_loaded_library": True, "static_mix": {"x": LONG}},
                     "hbm": {"algorithmic_bytes_per_launch": 14216640, "frac": 0.135}},
        "cpu_baseline": {"value": 46.03, "unit": "Mray/s", "cores": 16, "kind": "port", "sample": LONG, "ms_per_frame": 20.02,
                         "one_core_Mray_s": 2.99, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "interpreted": {"value": 0.0613},
                         "reference_itself": LONG},
        "boundary": {"value_at_c_abi": {"ms_per_frame": 0.2615, "note": LONG}, "python_hdrimage_note": LONG},
        "extra": {f"row{i}": {"note": LONG} for i in range(12)}, "frames_in_flight": {"note": LONG},
    }


def _canned_multi():
    row = {"value": 8900.0, "unit": "Mray/s", "ms_per_step": 0.51, "gather_check": "ok", "note": LONG}
    return {
        "metric": "Mray/s, C4 strong-scaled", "value": 30000.0, "unit": "Mray/s", "n_gpus": 8, "steps": 20, "warmup": 3, "ms_per_step": 0.15,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ranks_seen": 8,
        "backend": "nccl", "code_hash": "ab" * 32, "rays_traced_per_frame": 4580321,
        "config": {"workload": "C4 path tracer 3840x2160", "width": 3840, "height": 2160, "n_shapes": 257, "renderer": "PathTracer",
                   "pcg_mode": "PT_PCG_SAMPLE", "frames_in_flight_per_rank": 2, "partition": LONG},
        "gather": {"used": "whole", "probe_ms_per_frame": {"whole": 0.9, "sparse": None}, "gather_bytes_per_frame": 87091200,
                   "fallback_reason": "sparse failed: " + LONG, "rows_per_rank": list(range(8)), "note": LONG},
        "gather_bytes_per_frame_sent": 87091200, "gather_check": "ok", "without_gather": {"ms_per_step": 0.11},
        "oracle_check": {"checked": True, "bit_identical": True, "pixels_beyond_1e-5": 0, "rays_match": True, "against": LONG},
        "n1_same_workload": {"value": 8900.0, "ms_per_step": 0.51, "note": LONG}, "speedup": 3.4, "rank_share_imbalance": 1.02,
        "phases_ms": {"note": LONG, "render_ms": {"per_rank": [0.1] * 8}},
        "at_1280x720": dict(row, speedup=1.2, workload=LONG), "pcg_pixel": row, "c2_replicas": {"value": 4.9e5, "ms_per_step": 0.0149, "note": LONG},
        "value_note": LONG, "workload_note": LONG, "phase_seconds": [["x", 1.0]] * 40,
    }


def test_the_result_line_is_compact_and_carries_its_keys(tmp_path, monkeypatch, capsys):
    b = _bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    for full, compact_fn, need in ((_canned_n1(), b.compact_single, N1_KEYS),
                                   (_canned_multi(), b.compact_multi, {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                                       "higher_is_better", "scaling", "ranks_seen", "backend", "config",
                                                                       "gather", "parity_check", "n1_same_workload", "speedup", "detail_file"})):
        assert len(json.dumps(full)) > 100_000  # (the canned detail is as verbose as round 5's line was, and more)
        b.emit(full, compact_fn(full))
        out, err = capsys.readouterr()
        lines = [ln for ln in out.splitlines() if ln.strip()]
        line = lines[-1]
        assert len(lines) == 1 and len(line) < 4096 < b.COMPACT_LIMIT == 8192  # target 4 KB, hard limit 8 KB
        got = json.loads(line)
        assert need <= set(got), need - set(got)
        assert got["detail_file"] == b.DETAIL_NAME and json.load(open(tmp_path / b.DETAIL_NAME)) == full  # everything else: the side file ...
        assert err.startswith("[bench detail] ") and json.loads(err[len("[bench detail] "):]) == full       # ... and stderr
        if compact_fn is b.compact_single:
            assert ROOFLINE_KEYS <= set(got["roofline"]) and CPU_KEYS <= set(got["cpu_baseline"])
            assert set(got["config"]) == {"workload", "width", "height", "n_shapes", "renderer"}
            assert got["parity_check"] == {"bit_identical": True} and got["roofline"]["frac_bounds"] == [0.3307, 0.3396]
            assert got["ms_per_frame_at_c_abi"] == 0.2615 and got["roofline"]["algorithmic_bytes"] == 14216640
            assert got["value"] == 62103.0 and got["ms_per_step"] == 0.0148399  # six significant digits
        else:
            assert got["gather"]["used"] == "whole" and got["gather"]["fallback_reason"].startswith("sparse failed: ") and got["ranks_seen"] == 8
            assert got["parity_check"] == {"checked": True, "bit_identical": True, "pixels_beyond_1e-5": 0, "rays_match": True}


def test_a_line_that_would_not_fit_is_cut_not_printed_long():
    b = _bench()
    c = b.compact_single(_canned_n1())
    c["config"]["workload"] = "w" * 9000  # (cannot happen with the fixed key set; a line that does not parse loses the round)
    text = b.compact_dumps(c)
    assert len(text) < b.COMPACT_LIMIT and json.loads(text)["truncated"] is True and json.loads(text)["value"] == 62103.0
    assert json.loads(b.compact_dumps({"value": float("nan"), "x": [float("inf"), 1.0]})) == {"value": None, "x": [None, 1.0]}  # strict JSON


def test_the_headline_gather_falls_back_on_the_whole_shards():
    b = _bench()
    assert b.choose_gather(0.9, 0.4, None) == ("sparse", None)
    assert b.choose_gather(0.4, 0.9, None) == ("whole", None)
    used, why = b.choose_gather(0.9, None, "timed loop C4 sparse: RuntimeError: ncclInternalError")
    assert used == "whole" and why.startswith("sparse failed: timed loop C4 sparse")
    assert b.choose_gather(0.9, 0.4, "another rank failed")[0] == "whole"  # measured on this rank, failed on another: not used
    assert b.choose_gather(0.9, None, None) == ("whole", "sparse not measured")
    est, oracle_s = b.estimate_wall_s(8, 128)
    assert est < 8 * 60 and 0 < oracle_s < b.ORACLE_LIMIT_S
    est16, oracle16 = b.estimate_wall_s(8, 16)
    assert est16 < 8 * 60  # (16 cores: ~95 s of oracle, inside the limit; beyond the limit the check is skipped and costs nothing)


def test_a_phase_that_overruns_ends_the_job_with_the_line_measured_so_far(tmp_path, monkeypatch, capsys):
    """bench.py --gpus N: the watchdog's way out (a deadline passed, or the ranks can no longer talk).  Rank 0 prints the stashed
    whole-shard measurement marked with what happened and every rank leaves with status 0; with nothing stashed the line is an
    `error` line and the status 1.  (os._exit is replaced: the test process must survive.)"""
    b = _bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    left = []

    class Left(Exception):
        pass

    def fake_exit(code):
        left.append(code)
        raise Left()

    monkeypatch.setattr(b.os, "_exit", fake_exit)
    dog = b.Watchdog(0)
    dog.error_stub.update(n_gpus=8, steps=20, warmup=3, backend="nccl")
    import pytest

    with pytest.raises(Left):  # nothing measured yet: an error line, status 1
        dog.bail("phase 'warm-up C4 PT_PCG_SAMPLE whole gather' exceeded its deadline")
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert left == [1] and line["value"] is None and "exceeded its deadline" in line["error"] and line["n_gpus"] == 8
    dog2 = b.Watchdog(0)
    full = _canned_multi()
    full["gather"] = {"used": "whole", "probe_ms_per_frame": {"whole": 0.9}, "gather_bytes_per_frame": 87091200, "fallback_reason": "sparse not measured yet"}
    dog2.fallback, dog2.have_fallback = (full, None), True
    with pytest.raises(Left):
        dog2.bail("phase 'timed loop C4 PT_PCG_SAMPLE sparse gather gather=True' exceeded its deadline")
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert left == [1, 0] and line["value"] == 30000.0 and line["gather"]["used"] == "whole"
    assert "sparse gather gather=True' exceeded its deadline" in line["gather"]["fallback_reason"]
    assert line["comparable_with_the_n1_line"] is False and line["parity_check"]["bit_identical"] is True
    # a remote rank prints nothing and leaves with the same status (after a pause that lets rank 0 print first)
    monkeypatch.setattr(b.time, "sleep", lambda s: None)
    dog3 = b.Watchdog(3)
    dog3.have_fallback = True
    with pytest.raises(Left):
        dog3.bail("phase 'x': the ranks' agreement failed")
    assert left == [1, 0, 0] and capsys.readouterr().out == ""
