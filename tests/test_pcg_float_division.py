"""pcg_float on the device (csrc/pt_math.h) divides by 0xFFFFFFFF through the constant's reciprocal and one correction step;
tests/proofs/pcg_float_div.c proves by exhaustion -- all 2^32 inputs -- that the value is the IEEE quotient the reference
computes (pcg.py:60-62).  CPU only (gcc + OpenMP, ~5 s on 8 cores)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_the_reciprocal_form_is_the_ieee_quotient_for_all_2_to_32_inputs(tmp_path):
    exe = str(tmp_path / "pcg_float_div")
    r = subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "proofs", "pcg_float_div.c"), "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout
    assert "r = 0x1.00000001p-32" in run.stdout and "mismatches of the corrected quotient: 0;" in run.stdout
    # (and the correction step is needed: the plain product is wrong millions of times)
    assert "of the plain product: 5767168" in run.stdout


def test_the_kernel_source_uses_exactly_that_constant_and_sequence():
    src = open(os.path.join(os.path.dirname(HERE), "pytracer_amd", "csrc", "pt_math.h")).read()
    body = src[src.index("PT_DEV double pcg_unit(uint32_t v) {"):]  # (pcg_float = pcg_unit of the generator's next output)
    body = body[:body.index("\n}\n")]
    assert "PT_DEV double pcg_float(Pcg &p) { return pcg_unit(pcg_next(p)); }" in src
    assert "0x1.00000001p-32" in body and "__builtin_fma(-4294967295.0, q0, x)" in body and "__builtin_fma(e, r, q0)" in body
