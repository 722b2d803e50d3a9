#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in libccd_hip.so from the code object's metadata (no GPU needed).
    python tools/codeobj_regs.py [pattern]      -> one line per kernel: VGPRs (arch + accumulator), SGPRs, spills, scratch bytes"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ccd_amd", "libccd_hip.so")


def main():
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", SO, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    demangle = lambda n: subprocess.check_output(["c++filt", n], text=True).strip()
    rows = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        get = lambda k, blk=blk: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = re.sub(r"\(.*", "", demangle(get("name"))).replace("void ", "")
        agpr = blk.split("\n", 1)[0].strip()
        if pat and not pat.search(name):
            continue
        rows.append((name, get("vgpr_count"), agpr, get("sgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                     get("private_segment_fixed_size"), get("group_segment_fixed_size")))
    print("| kernel | VGPRs total (of which AGPRs) | SGPRs | VGPR spills | SGPR spills | scratch B | static LDS B |")
    print("|---|---|---|---|---|---|---|")
    for r in sorted(rows):
        print(f"| `{r[0]}` | {r[1]} ({r[2]}) | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} |")


if __name__ == "__main__":
    main()
