#!/usr/bin/env python3
"""Per-kernel SQ counter summary of a `rocprofv3 --pmc ...` pass over bench.py (tools/pmc_sq.sh).

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed
over SIMDs (32 per v_mfma_f32_32x32x16_bf16); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves, with WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES;
GRBM_GUI_ACTIVE is reported SUMMED over the 8 XCDs (cross-check: ccd::gemm_row384_kernel<7> runs 0.21 ms = 0.44 M cycles
and reports 3.96 M; its SQ_INSTS_MFMA = 2.95 M equals its algorithmic flops / 32768), so the busy time of a dispatch is
GUI_ACTIVE / 8 and  MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).
"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
SIMDS = 256 * 4
XCDS = 8


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:72]


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    count = collections.Counter()
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                count[k] += 1
    rows = []
    for k, c in per.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rows.append((gui, k, count[k], 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / XCDS * SIMDS),
                     100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
                     100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                     c.get("SQ_LDS_BANK_CONFLICT", 0) / max(count[k], 1)))
    total = sum(r[0] for r in rows)
    print("| kernel | dispatches | % of busy cycles | MFMA util % | waves parked % | issue-stalled % | issuing % | LDS-issuing % | "
          "LDS bank-conflict cycles / dispatch |")
    print("|---|---|---|---|---|---|---|---|---|")
    for gui, k, n, mfma, park, stall, act, lds, conf in sorted(rows, reverse=True):
        if gui / total < 0.002 or k.startswith("__amd_rocclr"):      # runtime copies of the start-up phase
            continue
        print(f"| `{k}` | {n} | {100 * gui / total:.1f} | {mfma:.1f} | {park:.1f} | {stall:.1f} | {act:.1f} | {lds:.1f} | {conf:.0f} |")


if __name__ == "__main__":
    main()
