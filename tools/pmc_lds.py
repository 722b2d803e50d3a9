#!/usr/bin/env python3
"""Per-kernel LDS / VALU counter summary of tools/pmc_lds.sh's `rocprofv3 --pmc` pass.

Units (MI355X_MICROARCH.md): SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra cycles among them (summed over
CUs); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_INST_* = quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs,
so a dispatch is busy GUI_ACTIVE / 8 shader cycles and
    LDS busy %     = LDS_IDX_ACTIVE / (GUI_ACTIVE / 8 * 256 CUs)
    conflict share = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE
"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
CUS, XCDS = 256, 8


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")[:72]


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
        busy = gui / XCDS
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        idx = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        rows.append((gui, k, count[k], 100 * idx / (busy * CUS), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(idx, 1.0),
                     c.get("SQ_LDS_ADDR_CONFLICT", 0) / max(count[k], 1), 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                     100 * c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_INSTS_VALU", 0) / max(count[k], 1),
                     c.get("SQ_INSTS_LDS", 0) / max(count[k], 1), c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_LDS", 0), 1.0)))
    total = sum(r[0] for r in rows)
    print("| kernel | dispatches | % of busy cycles | LDS array busy % of CU cycles | bank-conflict share of LDS cycles % | address-conflict cycles / dispatch | "
          "VALU-issuing % of wave cycles | LDS issue-stall % of wave cycles | VALU instr / dispatch | LDS instr / dispatch | VALU per LDS instr |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for gui, k, n, ldsb, conf, addr, valu, ldsw, nv, nl, ratio in sorted(rows, reverse=True):
        if gui / total < 0.002 or k.startswith("__amd_rocclr"):
            continue
        print(f"| `{k}` | {n} | {100 * gui / total:.1f} | {ldsb:.1f} | {conf:.1f} | {addr:.0f} | {valu:.1f} | {ldsw:.1f} | {nv:.3g} | {nl:.3g} | {ratio:.2f} |")


if __name__ == "__main__":
    main()
