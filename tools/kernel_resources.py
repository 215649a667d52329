#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of the engine's HIP sources (no GPU needed).

    python tools/kernel_resources.py [extra hipcc flags ...]

Compiles each .hip with -Rpass-analysis=kernel-resource-usage and prints one line per kernel.  The step kernels
must stay at 8 waves/SIMD (<= 64 VGPRs, <= 5 KB LDS per wave, no scratch): that is the first thing to check after
touching them.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contracts_amd import build as B  # noqa: E402


def resources(src, extra=()):
    cmd = [B.hipcc()] + B.FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c",
                                                 os.path.join(B.CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+))", line)
        if not m:
            continue
        if m.group(1):
            cur = {"name": m.group(1)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(2).strip()] = m.group(3)
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o.replace("void ce::", "")) for o in out]


def main():
    extra = sys.argv[1:]
    for src in B.SOURCES:
        rows = resources(src, extra)
        if not rows:
            continue
        names = demangle([r["name"] for r in rows])
        print("== %s" % src)
        print("%-44s %5s %5s %7s %7s %7s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "spills", "occ"))
        for r, nm in zip(rows, names):
            spills = "%s/%s" % (r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"))
            print("%-44s %5s %5s %7s %7s %7s %5s" % (nm[:44], r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                     r.get("ScratchSize", "?"), r.get("LDS Size", "?"), spills,
                                                     r.get("Occupancy", "?")))


if __name__ == "__main__":
    main()
