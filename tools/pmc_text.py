#!/usr/bin/env python3
"""profiles/pmc_*.json (tools/pmc_summary.py --json) -> the plain-text digest profiles/rNN_pmc_summary.txt.

    python3 tools/pmc_text.py profiles/pmc_c2.json profiles/pmc_c3_second_pass.json ... > profiles/r02_pmc_summary.txt
"""
import json
import os
import sys

for path in sys.argv[1:]:
    j = json.load(open(path))
    k = j["counters"]
    dur = j["dur_us"]
    print(f"{os.path.basename(path)[:-5]}: {j['kernel']}  launches {j['dispatches']}  duration under PMC collection {dur:.1f} us")
    print(f"   VALU wave-instructions {k['SQ_INSTS_VALU']:.4g}  SALU {k['SQ_INSTS_SALU']:.4g}  waves {k['SQ_WAVES']:.0f}")
    print(f"   VALU issue utilisation = instr x 4 cycles / (1024 SIMDs x duration x 2.4 GHz) = {k['SQ_INSTS_VALU'] * 4 / (1024 * dur * 2400):.3f}")
    if "SQ_WAIT_ANY" in k:
        print(f"   wave-cycles (quad-cycles) {k['SQ_WAVE_CYCLES']:.4g}: waiting on s_waitcnt/barrier {k['SQ_WAIT_ANY']:.4g}, issue stalls "
              f"{k['SQ_WAIT_INST_ANY']:.4g}, VALU active {k['SQ_ACTIVE_INST_VALU']:.4g}; the average wave is alive "
              f"{k['SQ_WAVE_CYCLES'] * 4 / k['SQ_WAVES'] / 2400 / dur:.2f} of the kernel's duration")
    if "hbm_bytes_per_launch" in j:
        print(f"   HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB counters) {j['hbm_bytes_per_launch']}")
    print(f"   source: {j['source']}")
    print()
