#!/usr/bin/env python3
"""Steady-state per-kernel summary of a `rocprofv3 --kernel-trace` run that wrote the default rocpd SQLite database
(`*_results.db`): aggregates the LAST `--steps` training iterations, cut at a marker kernel (first kernel of a step).

    python tools/prof_db_summary.py gpurun_out/prof_ft/ft_results.db --steps 2 --marker patch_embed_fwd > profiles/<name>.md
"""
import argparse
import collections
import re
import sqlite3


def short(name):
    name = name[:-3] if name.endswith(".kd") else name
    try:
        import subprocess
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--marker", default="patch_embed_fwd")
    ap.add_argument("--markers-per-step", type=int, default=1, help="launches of the marker kernel per training iteration "
                    "(pretraining: 2 - the student's and the teacher's patch embedding)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
    rows = sorted((s, e, names[k]) for s, e, k in cur.execute(f"select start, end, kernel_id from {kd}"))
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]][::a.markers_per_step]
    assert len(marks) >= a.steps + 1, f"only {len(marks)} steps in the trace"
    lo, hi = marks[-a.steps - 1], marks[-1]
    sel = rows[lo:hi]
    wall = (sel[-1][1] - sel[0][0]) / a.steps / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in sel:
        agg[short(n)][0] += 1
        agg[short(n)][1] += (e - s) / 1e6
    busy = sum(v[1] for v in agg.values()) / a.steps
    print(f"steady state over {a.steps} steps: wall {wall:.2f} ms/step, kernel-busy {busy:.2f} ms/step, "
          f"{len(sel) // a.steps} launches/step\n")
    print("| kernel | launches/step | ms/step | avg us | % |")
    print("|---|---|---|---|---|")
    for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {n} | {c / a.steps:.0f} | {ms / a.steps:.3f} | {1e3 * ms / c:.1f} | {100 * ms / a.steps / busy:.1f} |")


if __name__ == "__main__":
    main()
