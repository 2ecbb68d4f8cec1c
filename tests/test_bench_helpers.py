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
