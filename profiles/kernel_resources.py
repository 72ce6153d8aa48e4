#!/usr/bin/env python3
"""Register / LDS / spill table of the accumulate kernels, per kernel family and arithmetic contract, from the compiler's own
report (hipcc -Rpass-analysis=kernel-resource-usage; `make -C kiwi_amd/csrc asm FAMILY=n ARITH=exact|fused`).
Runs on the build machine (no GPU needed):   python profiles/kernel_resources.py [--json out.json]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kiwi_amd", "csrc")
FAMILIES = {1: "direct", 2: "grouped", 3: "multi", 4: "cell"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void kiwi::", "") for o in out]


def collect():
    rows = []
    for fam in FAMILIES:
        for ar in ("exact", "fused"):
            r = subprocess.run(["make", "-s", "-C", CSRC, "asm", "FAMILY=%d" % fam, "ARITH=%s" % ar], capture_output=True, text=True)
            txt = r.stderr + r.stdout
            cur = None
            for line in txt.split("\n"):
                m = re.search(r"Function Name: (\S+)", line)
                if m:
                    cur = {"family": FAMILIES[fam], "arith": ar, "mangled": m.group(1)}
                    rows.append(cur)
                    continue
                if cur is None:
                    continue
                for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                                 ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                                 ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"),
                                 ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                    m = re.search(pat, line)
                    if m:
                        cur[key] = int(m.group(1))
    names = demangle([r["mangled"] for r in rows])
    for r, n in zip(rows, names):
        r["kernel"] = n
        del r["mangled"]
    return rows


def main():
    rows = collect()
    print("| kernel | contract | VGPR | AGPR | SGPR | VGPR spills | SGPR spills | scratch B/lane | LDS B | waves/SIMD |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| `%s` | %s | %d | %d | %d | %d | %d | %d | %d | %d |" % (r["kernel"], r["arith"], r.get("vgpr", -1), r.get("agpr", 0), r.get("sgpr", -1),
                                                                       r.get("vgpr_spill", 0), r.get("sgpr_spill", 0), r.get("scratch", 0),
                                                                       r.get("lds", 0), r.get("occupancy", -1)))
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
