#!/usr/bin/env python3
"""Per-launch HBM traffic of the GEMM kinds from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, WRITE_SIZE).

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
  * both counters are reported in KiB -> x 1024;
  * on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads -> x 2;
  * WRITE_SIZE is uncalibrated -> calibrated here on a kernel of the SAME run whose byte counts are known exactly:
    ccd::ln_fwd_kernel over R x E fp32 rows reads 4*R*E B and writes 2*R*E + 8*R B (R = 131072, E = 384 at B = 256).
    The same kernel also cross-checks the FETCH correction (reported as fetch_check, expected 1.0).
Output: JSON {kind: {launches, fetch_bytes, write_bytes, bytes_per_launch}} for bench.py's `roofline.traffic`.
"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)
KINDS = {"<false, 0, false>": "gemm_nt_bf16", "<false, 1, false>": "gemm_nt_gelu", "<false, 2, false>": "gemm_nt_resid",
         "<false, 3, false>": "gemm_nt_f32", "<true, 4, false>": "gemm_tn_atomic", "<false, 5, false>": "gemm_nt_dgelu",
         "<false, 0, true>": "conv_gemm", "<true, 4, true>": "conv_wgrad", "<true, 3, false>": "gemm_tn_f32"}


def collect(path, counter):
    """{kind: [counter values of the dispatches of the LAST TWO training steps]}.  A step ends with its ccd::adamw_kernel dispatch:
    the run is initialisation (hundreds of small dispatches), one warm-up step, two timed steps - a cut by dispatch COUNT (what this
    file did until round 4) lands inside the initialisation and keeps all three steps."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append(r)
    ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    if len(ends) >= 3:
        rows = rows[ends[-3] + 1:ends[-1] + 1]
    elif len(ends) == 2:                                   # no warm-up step in the run: everything after the first step
        rows = rows[ends[0] + 1:ends[1] + 1] * 2
    per = collections.defaultdict(list)
    if True:
        for r in rows:
            name = r["Kernel_Name"]
            if "gemm_bf16_kernel" in name:
                m = re.search(r"gemm_bf16_kernel(<[^>]*>)", name)
                key = KINDS.get(m.group(1), m.group(1)) if m else "gemm?"
            elif "gemm256_kernel" in name:                   # same kinds as bench.py's KernelTimer keys
                m = re.search(r"gemm256_kernel<(\d+)", name)
                key = {"0": "gemm_nt_bf16", "1": "gemm_nt_gelu", "2": "gemm_nt_resid", "3": "gemm_nt_f32",
                       "5": "gemm_nt_dgelu"}.get(m.group(1), "gemm256?") if m else "gemm256?"
            elif "gemm_row384_kernel" in name:               # full-row kernel: residual (+ LayerNorm) products
                m = re.search(r"gemm_row384_kernel<(\d+)", name)
                key = {"0": "gemm_nt_bf16", "2": "gemm_nt_resid", "3": "gemm_nt_f32", "7": "gemm_nt_resid"}.get(
                    m.group(1), "gemm_row384?") if m else "gemm_row384?"
            elif "gemm_tn384_kernel" in name:                # weight-gradient products (single or paired) of the ViT blocks
                key = "gemm_tn_atomic"
            elif "mlp_fused_kernel" in name:                 # fc1 + GELU + fc2 + residual + LayerNorm in one launch; <E, store_u, true>: with
                key = "proj_mlp_fused" if re.search(r"mlp_fused_kernel<\d+, (true|false), true>", name) else "mlp_fused"   # the projection prologue (round 5)
            elif "rowproj_kernel" in name:                   # K = E bf16 projections with resident activation rows (round 3)
                key = "gemm_nt_bf16"
            elif "rowgemm_kernel" in name:                   # row-owner products with a row-wise epilogue
                m = re.search(r"rowgemm_kernel<\d+, \d+, (\d+)", name)
                key = "gemm_nt_lnbwd" if (m and m.group(1) == "0") else "gemm_nt_resid"
            elif "attention_fwd_kernel" in name:
                key = "attention_fwd"
            elif "attention_bwd_onepass_kernel" in name:     # round 4: one kernel per ops.attention_bwd call
                key = "attention_bwd"
            elif "attention_bwd_dq_kernel" in name:
                key = "attention_bwd_dq"
            elif "attention_bwd_dkv" in name:
                key = "attention_bwd_dkv"
            elif "ln_fwd_kernel" in name:
                key = "ln_fwd"
            else:
                key = None
            per["_all"].append(float(r["Counter_Value"]))    # every dispatch of the run: the step's total traffic
            if key is not None:
                per[key].append(float(r["Counter_Value"]))
    return per


def steady(vals):
    """(collect() already keeps the dispatches of the two timed steps only)"""
    return vals


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
R, E = 131072, 384
ln_read, ln_write = 4.0 * R * E, 2.0 * R * E + 8.0 * R
ln_f = sum(steady(fetch["ln_fwd"])) / len(steady(fetch["ln_fwd"])) * 1024 * 2
ln_w = sum(steady(write["ln_fwd"])) / len(steady(write["ln_fwd"])) * 1024
wcal = ln_write / ln_w
out = {"_method": {"fetch": "FETCH_SIZE[KiB] * 1024 * 2 (gfx950 128-B requests tallied as 64 B)",
                   "write": f"WRITE_SIZE[KiB] * 1024 * {wcal:.4f} (calibrated on ccd::ln_fwd_kernel, known {ln_write:.0f} B)",
                   "fetch_check": round(ln_f / ln_read, 4), "source": [sys.argv[1].split("gpurun_out/")[-1],
                                                                        sys.argv[2].split("gpurun_out/")[-1]]}}
# the whole step (the bytes budget): every dispatch of the two timed steps, the same corrections (the x 2 of
# FETCH_SIZE is exact for wide coalesced reads - what the step's large kernels issue - and an upper bound elsewhere)
_fa, _wa = steady(fetch["_all"]), steady(write["_all"])
out["_step_bytes"] = round((sum(_fa) * 1024 * 2 + sum(_wa) * 1024 * wcal) / 2)
out["_step_fetch_bytes"] = round(sum(_fa) * 1024 * 2 / 2)
out["_step_write_bytes"] = round(sum(_wa) * 1024 * wcal / 2)
for k in sorted(fetch):
    if k == "ln_fwd" or k == "_all":
        continue
    fv, wv = steady(fetch[k]), steady(write.get(k, [0.0]))
    fb = sum(fv) / len(fv) * 1024 * 2
    wb = sum(wv) / len(wv) * 1024 * wcal
    out[k] = {"launches": len(fv), "fetch_bytes": round(fb), "write_bytes": round(wb), "bytes_per_launch": round(fb + wb)}
if "attention_bwd_dq" in out and "attention_bwd_dkv" in out:      # one ops.attention_bwd call = the two kernels (+ a tiny finish kernel)
    a, b = out["attention_bwd_dq"], out["attention_bwd_dkv"]
    out["attention_bwd"] = {"launches": a["launches"], "fetch_bytes": a["fetch_bytes"] + b["fetch_bytes"],
                            "write_bytes": a["write_bytes"] + b["write_bytes"], "bytes_per_launch": a["bytes_per_launch"] + b["bytes_per_launch"]}
# identify the kernels these passes were taken on: bench.py reports `traffic` only while the digest still matches
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_digest
out["_kernel_sources_sha256"] = kernel_sources_digest()
print(json.dumps(out, indent=1))
