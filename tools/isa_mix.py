#!/usr/bin/env python3
"""What a kernel's VALU instructions ARE, instruction by instruction, and what they cost -- from the code object itself.

    python tools/isa_mix.py "pt_tile4_kernel<1, true>" [--lib libptrace.so] [--pmc profiles/pmc_c2.json [--update]]

rocprofv3's class counters (SQ_INSTS_VALU_ADD_F64, ..._INT32, ...) leave a third of pt_tile4_kernel's VALU instructions in no
class -- compares, selects, moves, lane reads, division helpers -- and bench.py priced those at a guessed 3 cycles (VERDICT r4
weak #2).  This tool removes the guess in two steps:

1. STATIC: the kernel's gfx950 disassembly (llvm-objdump of the code object in the library), every instruction put in a
   class with the SIMD-cycles tools/micro/issue.hip MEASURED for it (profiles/r03_issue_rates.txt, r03_issue_pmc_probe.txt);
   the few opcodes nobody measured are the only "unpriced" ones left and carry a [low, high] price.
2. DYNAMIC, bounded: how often each basic block runs is not known, but it is CONSTRAINED -- by the control-flow graph (what
   enters a block leaves it; the entry block runs once per wave: SQ_WAVES) and by every per-launch counter of the PMC summary
   (total VALU, the eleven VALU classes, LDS, VMEM, SMEM: each is a sum over blocks of (static count in the block) x (runs of
   the block)).  Two linear programmes over the edge flows give the MINIMUM and the MAXIMUM of the priced cycles that any
   execution consistent with all those measurements can have: `valu_issue_cycles_bounds`.  No block count is guessed.

With --pmc the bounds are computed against that file's counters; --update writes them back into it as `static_mix`
(bench.py: roofline.frac_bounds_from_disassembly).  Calls (s_swappc_b64 to pt_atan2d / pt_acosd ...) are followed: the callee's
blocks join the graph, entered as often as their call sites run.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kres  # noqa: E402

# ---- classes and prices ---------------------------------------------------------------------------------------------------
# (SIMD-cycles per wave64 instruction, measured: profiles/r03_issue_rates.txt, r03_issue_pmc_probe.txt (tools/micro/issue.hip)
#  and profiles/r05_issue2_rates.txt (tools/micro/issue2.hip: the opcodes the kernels' disassembly holds that round 3 had not
#  measured -- division helpers, ldexp / floor / trunc, class and integer compares, 64-bit moves, min / max, shifts, lane
#  writes: ALL of them issue like v_add_f64, 4 cycles; only add / sub / mul / fma / and / or / xor / mov on 32 bits take 2).
#  A (low, high) pair = not measured.)
TABLE = [  # (regex on the mnemonic, class, cycles or (low, high), rocprofv3 class counter it is believed to land in or None)
    (r"v_add_f64|v_sub_f64", "add_f64", 4, "SQ_INSTS_VALU_ADD_F64"),
    (r"v_mul_f64", "mul_f64", 4, "SQ_INSTS_VALU_MUL_F64"),
    (r"v_fma_f64|v_fmac_f64", "fma_f64", 4, "SQ_INSTS_VALU_FMA_F64"),
    (r"v_(rcp|rsq|sqrt)_f64", "trans_f64", 16, "SQ_INSTS_VALU_TRANS_F64"),
    (r"v_div_(scale|fmas|fixup)_f64|v_ldexp_f64|v_(floor|trunc|ceil|rndne|fract)_f64|v_frexp_(mant|exp_i32)_f64|v_(min|max)_f64", "misc_f64", 4, None),
    (r"v_trig_preop_f64", "trig_preop_f64", (4, 16), None),  # (ocml's large-argument reduction; not measured)
    (r"v_cmp[x]?_class_f64", "cmp", 4, None),
    # (integer compares and 32-bit bit operations land in SQ_INSTS_VALU_INT32: calibrated -- of nine candidate mappings of that
    #  counter only "add/sub/mul + bit operations + 32-bit integer compares" leaves the programme feasible on pt_tile4_kernel)
    (r"v_cmp[x]?_\w+_(i32|u32)", "cmp_int32", 4, "SQ_INSTS_VALU_INT32"),
    (r"v_cmp[x]?_\w+_(f64|f32|f16|i64|u64|i16|u16)", "cmp", 4, None),
    (r"v_cndmask_b32", "cndmask", 4, None),
    (r"v_(readlane|readfirstlane|writelane)_b32|v_mbcnt_(lo|hi)_u32_b32", "lane", 4, None),
    (r"v_pk_(mul|add|fma)_f32", "pk_f32", 4, None),
    (r"v_pk_mov_b32", "mov", (2, 4), None),
    (r"v_(add|sub|subrev)_f32", "add_f32", 2, "SQ_INSTS_VALU_ADD_F32"),
    (r"v_mul_f32|v_mul_legacy_f32", "mul_f32", 2, "SQ_INSTS_VALU_MUL_F32"),
    (r"v_(fma|fmac|mad|mac)_f32", "fma_f32", 2, "SQ_INSTS_VALU_FMA_F32"),
    (r"v_(rcp|rcp_iflag|rsq|sqrt|exp|log|sin|cos)_f32", "trans_f32", 8, "SQ_INSTS_VALU_TRANS_F32"),
    (r"v_(min|max|min3|max3|med3)_f32|v_ldexp_f32|v_(floor|trunc|ceil|rndne|fract)_f32|v_frexp_\w+_f32", "misc_f32", 4, None),
    (r"v_cvt_\w+", "cvt", 4, "SQ_INSTS_VALU_CVT"),
    (r"v_mov_b64", "mov64", 4, None),
    (r"v_mov_b32|v_accvgpr_\w+", "mov", 2, None),
    (r"v_(lshlrev|lshrrev|ashrrev)_[bi]64|v_lshl_add_u64|v_mad_[ui]64_[ui]32|v_(add|sub)_(co_)?[ui]64", "int64", 4, "SQ_INSTS_VALU_INT64"),
    (r"v_mul_(lo|hi)_[ui]32|v_mul_[ui]32_[ui]24|v_mad_[ui]32_[ui]24", "mul_int32", 4, "SQ_INSTS_VALU_INT32"),
    (r"v_(add|sub|subrev|addc|subb|subbrev)(_co)?_[ui]32|v_add3_u32|v_(add_lshl|lshl_add|lshl_or|and_or|or3|xad)_[ub]32|"
     r"v_(min|max|min3|max3|med3)_[ui]32", "int32", 2, "SQ_INSTS_VALU_INT32"),
    (r"v_(and|or|xor)_b32", "logic32", 2, "SQ_INSTS_VALU_INT32"),
    (r"v_(lshlrev|lshrrev|ashrrev)_[bi]32|v_bfe_[ui]32", "shift32", 4, "SQ_INSTS_VALU_INT32"),
    (r"v_(not|bfi|bfm|bfrev|bitop3|alignbit|alignbyte|perm|ffbh|ffbl|bcnt)_\w+", "bit32", (2, 4), "SQ_INSTS_VALU_INT32"),
]
TABLE = [(re.compile("^(?:" + pat + ")(?:_e32|_e64|_sdwa|_dpp)?$"), cls, cyc, ctr) for pat, cls, cyc, ctr in TABLE]
VALU_CLASS_COUNTERS = sorted({ctr for _, _, _, ctr in TABLE if ctr})


def classify(mn):
    """-> (unit, class, (low, high) cycles, counter) for a mnemonic."""
    if mn.startswith("v_"):
        for rx, cls, cyc, ctr in TABLE:
            if rx.match(mn):
                lo, hi = cyc if isinstance(cyc, tuple) else (cyc, cyc)
                return "valu", cls, (lo, hi), ctr
        return "valu", "unknown:" + mn, (2, 16), None
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_buffer_store", "s_dcache", "s_memtime", "s_memrealtime", "s_atc_probe")):
        return "smem", "smem", (0, 0), None
    if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch", "branch", (0, 0), None
    if mn.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_sethalt", "s_setprio", "s_sendmsg", "s_trap",
                      "s_icache", "s_incperflevel", "s_decperflevel", "s_ttrace", "s_set_gpr_idx", "s_code_end")):
        return "misc", "misc", (0, 0), None
    if mn.startswith("s_"):
        return "salu", "salu", (0, 0), None
    if mn.startswith("ds_"):
        return "lds", "lds", (0, 0), None
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_")):
        return "vmem", "vmem", (0, 0), None
    return "other", "other:" + mn, (0, 0), None


# ---- disassembly -> functions -> basic blocks --------------------------------------------------------------------------------
LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
HEAD = re.compile(r"^([0-9a-f]+) <(\S+)>:")
TARGET = re.compile(r"<\S+?\+0x([0-9a-fA-F]+)>\s*$")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        co = kres.code_object(lib, os.path.join(d, "gfx950.co"))
        txt = subprocess.run([f"{kres.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for ln in txt.splitlines():
        h = HEAD.match(ln)
        if h:
            cur = {"name": h.group(2), "start": int(h.group(1), 16), "insns": []}
            funcs[cur["name"]] = cur
            continue
        m = LINE.match(ln)
        if m and cur is not None:
            t = TARGET.search(ln)
            cur["insns"].append({"mn": m.group(1), "ops": m.group(2), "addr": int(m.group(3), 16),
                                 "target": cur["start"] + int(t.group(1), 16) if t else None})
    return funcs


def call_target(insns, k):
    """insns[k] is s_swappc_b64 sA, s[x:y]: find `s_getpc_b64 s[x:y]` + `s_add_u32 sx, sx, LIT` before it -> absolute target."""
    reg = insns[k]["ops"].split(",")[-1].strip()
    lo = reg[2:].split(":")[0] if reg.startswith("s[") else None
    for j in range(k - 1, max(-1, k - 200), -1):
        if insns[j]["mn"] == "s_getpc_b64" and insns[j]["ops"].strip() == reg:
            for i in range(j + 1, min(k, j + 6)):
                if insns[i]["mn"] == "s_add_u32" and insns[i]["ops"].split(",")[0].strip() == f"s{lo}":
                    lit = int(insns[i]["ops"].split(",")[-1].strip(), 0)
                    if lit >= 1 << 31:
                        lit -= 1 << 32
                    return insns[j]["addr"] + 4 + lit
    return None


def blocks_of(func, by_addr):
    """Basic blocks of one function -> list of {"insns", "succ": [addr...], "calls": [addr...], "exit": bool}."""
    insns = func["insns"]
    leaders = {insns[0]["addr"]}
    for k, ins in enumerate(insns):
        unit = classify(ins["mn"])[0]
        if unit == "branch" or ins["mn"] == "s_endpgm":
            if ins["target"] is not None:
                leaders.add(ins["target"])
            if k + 1 < len(insns):
                leaders.add(insns[k + 1]["addr"])
    blocks, cur = [], None
    for k, ins in enumerate(insns):
        if ins["addr"] in leaders:
            cur = {"addr": ins["addr"], "insns": [], "succ": [], "calls": [], "exit": False, "func": func["name"]}
            blocks.append(cur)
        cur["insns"].append(ins)
        nxt = insns[k + 1]["addr"] if k + 1 < len(insns) else None
        last = nxt is None or nxt in leaders
        if not last:
            continue
        mn = ins["mn"]
        if mn == "s_endpgm" or mn.startswith("s_setpc"):
            cur["exit"] = True
        elif mn == "s_branch":
            cur["succ"] = [ins["target"]]
        elif mn.startswith("s_cbranch"):
            cur["succ"] = [ins["target"]] + ([nxt] if nxt is not None else [])
        elif mn.startswith("s_swappc"):
            tgt = call_target(insns, k)
            if tgt is None or tgt not in by_addr:
                raise SystemExit(f"{func['name']}: cannot resolve the call at {ins['addr']:#x}")
            cur["calls"] = [tgt]
            cur["succ"] = [nxt] if nxt is not None else []
        elif nxt is not None:
            cur["succ"] = [nxt]
    return blocks


def build_graph(funcs, kernel):
    by_addr = {f["start"]: f for f in funcs.values()}
    todo, seen, blocks = [kernel], set(), []
    while todo:
        name = todo.pop()
        if name in seen:
            continue
        seen.add(name)
        bs = blocks_of(funcs[name], by_addr)
        blocks += bs
        for b in bs:
            for c in b["calls"]:
                todo.append(by_addr[c]["name"])
    return blocks, by_addr


def static_counts(blocks):
    rows = []
    for b in blocks:
        c = Counter()
        lo = hi = 0
        for ins in b["insns"]:
            unit, cls, (l, h), ctr = classify(ins["mn"])
            c["unit:" + unit] += 1
            if unit == "valu":
                c["cls:" + cls] += 1
                if ctr:
                    c["ctr:" + ctr] += 1
                lo += l
                hi += h
        rows.append({"n": c, "lo": lo, "hi": hi})
    return rows


def lp_bounds(blocks, rows, entry_addr, by_addr, counters, tol, use):
    """min / max of sum_b runs_b * cost_b over every edge flow consistent with the graph and the counters."""
    import numpy as np
    from scipy.optimize import linprog
    from scipy.sparse import lil_matrix

    idx = {b["addr"]: i for i, b in enumerate(blocks)}
    edges = []  # (from block index or -1 = source, to block index)
    for i, b in enumerate(blocks):
        for s in b["succ"]:
            if s in idx:
                edges.append((i, idx[s]))
        for c in b["calls"]:
            edges.append((i, idx[c]))  # a call edge: carries the call site's runs into the callee (not taken from the caller's flow)
    edges.append((-1, idx[entry_addr]))
    ne, nb = len(edges), len(blocks)
    call_edge = {(i, idx[c]) for i, b in enumerate(blocks) for c in b["calls"]}
    # runs_b = sum of in-flows (call edges included: a callee's entry is entered by its call sites)
    inflow = lil_matrix((nb, ne))
    outflow = lil_matrix((nb, ne))
    for e, (u, v) in enumerate(edges):
        inflow[v, e] = 1.0
        if u >= 0 and (u, v) not in call_edge:
            outflow[u, e] = 1.0
    A_eq, b_eq = [], []
    for i, b in enumerate(blocks):
        if not b["exit"]:
            A_eq.append((inflow[i] - outflow[i]).toarray()[0])  # what enters leaves
            b_eq.append(0.0)
        for c in b["calls"]:  # the call edge carries exactly the call site's runs
            row = inflow[i].toarray()[0].copy()
            row[edges.index((i, idx[c]))] -= 1.0
            A_eq.append(row)
            b_eq.append(0.0)
    src = np.zeros(ne)
    src[ne - 1] = 1.0
    A_eq.append(src)
    b_eq.append(float(counters["SQ_WAVES"]))
    A_ub, b_ub = [], []
    inflow_d = inflow.toarray()

    def constrain(per_block, value):
        row = np.array(per_block, dtype=float) @ inflow_d
        A_ub.append(row)
        b_ub.append(value * (1.0 + tol))
        A_ub.append(-row)
        b_ub.append(-value * (1.0 - tol))

    for name, key in use:
        if name in counters:
            constrain([r["n"][key] for r in rows], float(counters[name]))
    out = {}
    for tag, cost, sign in (("min", [r["lo"] for r in rows], 1.0), ("max", [r["hi"] for r in rows], -1.0)):
        c = sign * (np.array(cost, dtype=float) @ inflow_d)
        res = linprog(c, A_ub=np.array(A_ub), b_ub=np.array(b_ub), A_eq=np.array(A_eq), b_eq=np.array(b_eq), bounds=(0, None), method="highs")
        if res.status != 0:
            return None, res.message
        out[tag] = sign * res.fun
        out[tag + "_runs"] = inflow_d @ res.x
    return out, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", help="demangled kernel name or a substring of it")
    ap.add_argument("--lib", default=os.environ.get("PTRACE_LIB") or kres.DEFAULT_LIB)
    ap.add_argument("--pmc", help="PMC summary (tools/pmc_summary.py --json) with the kernel's per-launch counters")
    ap.add_argument("--update", action="store_true", help="write the result into the --pmc file as `static_mix`")
    ap.add_argument("--tol", type=float, default=0.002, help="relative slack on every counter (separate passes, medians)")
    args = ap.parse_args()

    funcs = disassemble(args.lib)
    dem = kres.demangle(list(funcs))
    want = args.kernel.replace("void ", "").strip()
    match = [m for m, d in dem.items() if d.replace("void ", "").split("(")[0].strip() == want.split("(")[0].strip()] or \
            [m for m, d in dem.items() if want in d]
    if len(match) != 1:
        raise SystemExit(f"{len(match)} kernels match {args.kernel!r}: {[dem[m] for m in match][:8]}")
    kernel = match[0]
    blocks, by_addr = build_graph(funcs, kernel)
    rows = static_counts(blocks)

    total = Counter()
    for r in rows:
        total.update(r["n"])
    n_valu = total["unit:valu"]
    print(f"{dem[kernel]}: {sum(len(b['insns']) for b in blocks)} instructions in {len(blocks)} basic blocks "
          f"({len({b['func'] for b in blocks})} functions), {n_valu} VALU")
    print("static VALU mix (instructions in the code, NOT weighted by how often they run):")
    for k, v in sorted(((k[4:], v) for k, v in total.items() if k.startswith("cls:")), key=lambda kv: -kv[1]):
        print(f"  {k:24s} {v:6d}  {100.0 * v / n_valu:5.1f} %")
    unknown = {k[4:]: v for k, v in total.items() if k.startswith("cls:unknown")}
    ranged = sum(1 for b in blocks for ins in b["insns"] if classify(ins["mn"])[0] == "valu" and classify(ins["mn"])[2][0] != classify(ins["mn"])[2][1])
    print(f"opcodes without a measured price: {ranged} of {n_valu} VALU instructions ({100.0 * ranged / n_valu:.1f} % static){' ' + str(unknown) if unknown else ''}")
    result = {"kernel": dem[kernel], "basic_blocks": len(blocks), "valu_instructions_static": n_valu,
              "static_valu_mix": {k[4:]: v for k, v in total.items() if k.startswith("cls:")},
              "static_share_without_measured_price": ranged / n_valu}
    try:
        from pytracer_amd.build import code_hash
    except ImportError:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from pytracer_amd.build import code_hash
    result["code_hash"] = code_hash(args.lib)

    if args.pmc:
        pmc = json.load(open(args.pmc))
        if pmc.get("code_hash") and pmc["code_hash"] != result["code_hash"]:
            raise SystemExit(f"{args.pmc} was collected on device code {pmc['code_hash'][:16]}, {args.lib} is {result['code_hash'][:16]}")
        counters = pmc["counters"]
        # which counters the model can use: the units, then the VALU classes whose opcode mapping is believed -- tried one by
        # one, a class whose constraint makes the programme infeasible is reported and left out (the mapping, not the hardware,
        # is then in doubt)
        base = [("SQ_INSTS_VALU", "unit:valu"), ("SQ_INSTS_LDS", "unit:lds"), ("SQ_INSTS_VMEM", "unit:vmem"), ("SQ_INSTS_SMEM", "unit:smem")]
        entry = funcs[kernel]["start"]
        use, dropped = [], []
        for cand in base + [(c, "ctr:" + c) for c in VALU_CLASS_COUNTERS]:
            got, err = lp_bounds(blocks, rows, entry, by_addr, counters, args.tol, use + [cand])
            if got is None:
                dropped.append(cand[0])
            else:
                use.append(cand)
        got, err = lp_bounds(blocks, rows, entry, by_addr, counters, args.tol, use)
        if got is None:
            raise SystemExit(f"the programme is infeasible even without class counters: {err}")
        print(f"counters used as constraints: {[u[0] for u in use]}")
        if dropped:
            print(f"counters LEFT OUT (infeasible with this opcode mapping): {dropped}")
        valu = counters["SQ_INSTS_VALU"]
        print(f"priced VALU issue cycles per launch, over every execution consistent with the graph and the counters: "
              f"[{got['min']:.4g}, {got['max']:.4g}]  = [{got['min'] / valu:.3f}, {got['max'] / valu:.3f}] cycles per VALU instruction")
        # the dynamic mix at the two extremes (how the unclassed instructions split)
        mixes = {}
        for tag in ("min", "max"):
            runs = got[tag + "_runs"]
            mix = Counter()
            for r, x in zip(rows, runs):
                for k, v in r["n"].items():
                    if k.startswith("cls:"):
                        mix[k[4:]] += v * x
            mixes[tag] = {k: round(v) for k, v in sorted(mix.items(), key=lambda kv: -kv[1])}
        classed = sum(counters.get(c, 0.0) for c in VALU_CLASS_COUNTERS)
        print(f"rocprofv3 leaves {valu - classed:.0f} of {valu:.0f} VALU instructions in no class; the disassembly names them "
              f"(dynamic count at the cheapest / dearest consistent execution):")
        hw = {cls for _, cls, _, ctr in TABLE if ctr}
        for k in sorted(set(mixes["min"]) | set(mixes["max"]), key=lambda k: -mixes["max"].get(k, 0)):
            if k not in hw:
                print(f"  {k:24s} {mixes['min'].get(k, 0):9d} .. {mixes['max'].get(k, 0):9d}")
        result.update({"counters_used": [u[0] for u in use], "counters_left_out": dropped, "tolerance": args.tol,
                       "valu_issue_cycles_bounds": [got["min"], got["max"]],
                       "cycles_per_valu_instruction_bounds": [got["min"] / valu, got["max"] / valu],
                       "dynamic_mix_at_min": mixes["min"], "dynamic_mix_at_max": mixes["max"],
                       "note": "tools/isa_mix.py: every VALU instruction of the kernel's disassembly priced with its measured SIMD-cycles; "
                               "block execution counts bounded by two linear programmes over the control-flow graph and the per-launch "
                               "counters of this file (no block count guessed)"})
        if args.update:
            pmc["static_mix"] = result
            with open(args.pmc, "w") as f:
                json.dump(pmc, f, indent=1)
            print(f"static_mix written into {args.pmc}")
    else:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
