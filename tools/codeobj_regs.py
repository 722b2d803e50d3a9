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


def load(so=SO):
    """-> {demangled kernel name without arguments: dict(vgpr (architectural + accumulator, as allocated), agpr, sgpr, vgpr_spill, sgpr_spill,
    scratch, lds)} from the gfx950 code object inside `so`."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    blocks = notes.split("  - .agpr_count:")[1:]
    mangled = [(re.search(r"\.name:\s*(\S+)", b) or [None, "?"])[1] for b in blocks]
    names = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
    out = {}
    for blk, name in zip(blocks, names):
        get = lambda k, blk=blk: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        out[re.sub(r"\(.*", "", name).replace("void ", "")] = dict(
            vgpr=get("vgpr_count"), agpr=blk.split("\n", 1)[0].strip(), sgpr=get("sgpr_count"), vgpr_spill=get("vgpr_spill_count"),
            sgpr_spill=get("sgpr_spill_count"), scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))
    return out


def main():
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    print("| kernel | VGPRs total (of which AGPRs) | SGPRs | VGPR spills | SGPR spills | scratch B | static LDS B |")
    print("|---|---|---|---|---|---|---|")
    for name, r in sorted(load().items()):
        if pat and not pat.search(name):
            continue
        print(f"| `{name}` | {r['vgpr']} ({r['agpr']}) | {r['sgpr']} | {r['vgpr_spill']} | {r['sgpr_spill']} | {r['scratch']} | {r['lds']} |")


if __name__ == "__main__":
    main()
