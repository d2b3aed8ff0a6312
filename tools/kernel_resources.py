#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of one .hip file of the library (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py abz_smc_swarm.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "abcdez.jl_amd", "csrc")


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-Ibuild", "-I../../include", "--cuda-device-only", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print("%-90s %5s %5s %5s %6s %7s %5s" % ("kernel", "VGPR", "SGPR", "occ", "LDS", "scratch", "spill"))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        print("%-90s %5s %5s %5s %6s %7s %5s" % (name[:90], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("Occupancy [waves/SIMD]"),
                                                  r.get("LDS Size [bytes/block]"), r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill")))


if __name__ == "__main__":
    main()
