"""tools/isa_mix.py -- the roofline's price list is the kernel's own disassembly (no GPU: llvm-objdump of the code object in
the built library, scipy's linear programmes).

* every opcode class has the price tools/micro/issue.hip / issue2.hip measured; what carries a (low, high) pair is what
  nobody measured: under 1 % of the headline kernel's VALU instructions;
* the control-flow graph of the headline kernel is whole: calls resolved (pt_atan2d / pt_acosd), every block reachable;
* and when profiles/pmc_c2.json was collected on THIS build (its code_hash says so), the bounds it records are reproduced from
  its own counters -- the number in the bench line can be recomputed from profiles/ alone.
"""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LLVM = "/opt/rocm/lib/llvm/bin"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")) or shutil.which("c++filt") is None,
                                reason="needs the ROCm llvm tools")
KERNEL = "pt_tile4_kernel<1, true>"


def test_classes_and_prices():
    import isa_mix as im

    assert im.classify("v_add_f64")[1:3] == ("add_f64", (4, 4)) and im.classify("v_fmac_f64_e32")[3] == "SQ_INSTS_VALU_FMA_F64"
    assert im.classify("v_add_f32_e32")[2] == (2, 2) and im.classify("v_max_f32_e32")[2] == (4, 4)  # (issue2: min / max issue like fp64)
    assert im.classify("v_rcp_f64_e32")[2] == (16, 16) and im.classify("v_sqrt_f32_e32")[2] == (8, 8)
    assert im.classify("v_cmp_lt_i32_e32")[3] == "SQ_INSTS_VALU_INT32" and im.classify("v_cmp_lt_f64_e32")[3] is None
    assert im.classify("v_cndmask_b32_e32")[1:3] == ("cndmask", (4, 4)) and im.classify("v_mov_b64_e32")[2] == (4, 4)
    assert im.classify("v_lshlrev_b32_e32")[2] == (4, 4) and im.classify("v_and_b32_e32")[2] == (2, 2)
    assert im.classify("s_load_dwordx8")[0] == "smem" and im.classify("s_cbranch_execz")[0] == "branch"
    assert im.classify("ds_read_b64")[0] == "lds" and im.classify("global_store_dwordx3")[0] == "vmem" and im.classify("s_and_b64")[0] == "salu"
    unknown = im.classify("v_some_future_opcode")
    assert unknown[1].startswith("unknown:") and unknown[2][0] < unknown[2][1]


@pytest.fixture(scope="module")
def graph():
    import isa_mix as im
    import kres

    funcs = im.disassemble(kres.DEFAULT_LIB)
    dem = kres.demangle(list(funcs))
    kernel = [m for m, d in dem.items() if KERNEL in d]
    assert len(kernel) == 1
    blocks, by_addr = im.build_graph(funcs, kernel[0])
    return im, funcs, kernel[0], blocks, by_addr


def test_headline_kernel_graph_is_whole(graph):
    im, funcs, kernel, blocks, by_addr = graph
    addrs = {b["addr"] for b in blocks}
    assert funcs[kernel]["start"] in addrs and len({b["func"] for b in blocks}) >= 3  # the kernel + pt_atan2d + pt_acosd
    for b in blocks:
        padding = all(ins["mn"] in ("s_nop", "s_code_end") for ins in b["insns"])  # (behind a function's last s_setpc / s_endpgm)
        assert b["exit"] or b["succ"] or padding, hex(b["addr"])
        assert all(s in addrs for s in b["succ"]) and all(c in addrs for c in b["calls"])
    # reachable from the entry (through successors and calls)
    seen, todo = set(), [funcs[kernel]["start"]]
    by = {b["addr"]: b for b in blocks}
    while todo:
        a = todo.pop()
        if a in seen:
            continue
        seen.add(a)
        todo += by[a]["succ"] + by[a]["calls"]
    assert len(seen) >= 0.98 * len(blocks)  # (alignment padding after an s_endpgm may form a dead block)
    rows = im.static_counts(blocks)
    n_valu = sum(r["n"]["unit:valu"] for r in rows)
    unpriced = sum(1 for b in blocks for ins in b["insns"] if im.classify(ins["mn"])[0] == "valu" and im.classify(ins["mn"])[2][0] != im.classify(ins["mn"])[2][1])
    assert n_valu > 4000 and unpriced / n_valu < 0.01
    assert not any(k.startswith("cls:unknown") for r in rows for k in r["n"])


def test_the_committed_profile_reproduces_its_own_bounds(graph):
    from pytracer_amd.build import code_hash

    im, funcs, kernel, blocks, by_addr = graph
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_c2.json")))
    if pmc.get("code_hash") != code_hash():
        pytest.skip("profiles/pmc_c2.json was collected on another build of the library (bench.py then prices nothing from it)")
    mix = pmc["static_mix"]
    rows = im.static_counts(blocks)
    use = [(c, "unit:" + u) for c, u in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_VMEM", "vmem"), ("SQ_INSTS_SMEM", "smem"))]
    use += [(c, "ctr:" + c) for c in mix["counters_used"] if c.startswith("SQ_INSTS_VALU_")]
    got, err = im.lp_bounds(blocks, rows, funcs[kernel]["start"], by_addr, pmc["counters"], mix["tolerance"], use)
    assert got is not None, err
    lo, hi = mix["valu_issue_cycles_bounds"]
    assert abs(got["min"] - lo) <= 1e-6 * lo and abs(got["max"] - hi) <= 1e-6 * hi
    assert 3.0 < lo / pmc["counters"]["SQ_INSTS_VALU"] < hi / pmc["counters"]["SQ_INSTS_VALU"] < 4.2 and (hi - lo) / lo < 0.06
    assert mix["counters_left_out"] == []
