#!/usr/bin/env python3
"""Steady-state per-kernel summary of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

The whole-process `--stats` table is dominated by one-off work (MIOpen's solver search for the segmentation-head
convolutions launches naive reference kernels during the warm-up steps), so this script cuts the trace at the
student patch-embedding launches and aggregates the LAST `--steps` training iterations only.

    python tools/prof_summary.py gpurun_out/prof_r1b/bench_kernel_trace.csv --steps 2 > profiles/<name>.md
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--sequence", default=None, help="also write the LAST iteration launch by launch (markdown) to this file")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            grid = int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            wg = int(r.get("Workgroup_Size_X", 1) or 1) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["VGPR_Count"]),
                         int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]), grid // max(wg, 1), wg))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "patch_embed_fwd_kernel" in r[2]]
    assert len(marks) >= 2 * a.steps, "trace too short"
    if a.sequence:
        # the iteration BEFORE the last one (the last one's tail runs into the end of the process: readbacks, idle queue), up to and
        # including the first launch of the next, so that the gap at the iteration boundary shows
        last = rows[marks[-4]:marks[-2] + 1] if len(marks) >= 4 else rows[marks[-2]:]
        with open(a.sequence, "w") as f:
            f.write(f"# one steady-state iteration of `{a.trace}`, launch by launch ({len(last)} launches, the last row = the next iteration's first)\n\n")
            f.write("| # | start ms | kernel | workgroups x threads | us | gap before us |\n|---|---|---|---|---|---|\n")
            t0, prev_end = last[0][0], last[0][0]
            for i, (s_, e_, n, v, av, lds, wgs, wg) in enumerate(last):
                f.write(f"| {i} | {(s_ - t0) / 1e6:.3f} | `{short(n)[:70]}` | {wgs} x {wg} | {(e_ - s_) / 1e3:.1f} | {(s_ - prev_end) / 1e3:.1f} |\n")
                prev_end = max(prev_end, e_)
    rows = [r[:6] for r in rows]
    begin = marks[-2 * a.steps]                      # two launches per iteration: student, then teacher
    sel = rows[begin:]
    wall_ms = (sel[-1][1] - sel[0][0]) / 1e6
    agg = collections.OrderedDict()
    for s, e, n, v, av, lds in sel:
        d = agg.setdefault(short(n), {"calls": 0, "ns": 0, "vgpr": v, "agpr": av, "lds": lds, "min": 1 << 62, "max": 0})
        d["calls"] += 1
        d["ns"] += e - s
        d["min"], d["max"] = min(d["min"], e - s), max(d["max"], e - s)
    busy_ms = sum(d["ns"] for d in agg.values()) / 1e6
    print(f"# steady-state kernel summary: last {a.steps} iteration(s) of `{a.trace}`\n")
    print(f"wall {wall_ms / a.steps:.3f} ms/iteration, kernel-busy {busy_ms / a.steps:.3f} ms/iteration "
          f"({100 * busy_ms / wall_ms:.1f} % of wall)\n")
    # registers: the trace's VGPR_Count / Accum_VGPR_Count columns read 256 + 0 for a kernel that holds 512 (256 of them accumulator
    # registers) - the code object's own metadata is what the table shows where the library is at hand (tools/codeobj_regs.py)
    try:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import codeobj_regs
        meta = codeobj_regs.load()
    except Exception:
        meta = {}
    print("| kernel | calls/iter | total ms/iter | % busy | avg us | min us | max us | registers: total (accumulator) | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        m = meta.get(k)
        regs = f"{m['vgpr']} ({m['agpr']})" if m else f"{d['vgpr']}+{d['agpr']} (trace)"
        print(f"| `{k}` | {d['calls'] / a.steps:.1f} | {d['ns'] / 1e6 / a.steps:.3f} | {100 * d['ns'] / 1e6 / busy_ms:.1f} | "
              f"{d['ns'] / d['calls'] / 1e3:.1f} | {d['min'] / 1e3:.1f} | {d['max'] / 1e3:.1f} | {regs} | {d['lds']} |")


if __name__ == "__main__":
    main()
