#!/usr/bin/env python
"""Per-kernel register / scratch / LDS / occupancy table of one .hip unit (hipcc -Rpass-analysis=kernel-resource-usage;
cross-compiles for gfx950 without a GPU).  usage: python tools/kernel_resources.py binning [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "latentsplat_amd", "csrc")
FLAGS = {"preprocess": ["-ffp-contract=off"], "sh": ["-ffp-contract=off"]}   # per-unit flags of csrc/Makefile (COMMON is spelled out below)


def main():
    unit = sys.argv[1]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I.", "-fno-fast-math", "-fno-slp-vectorize",
           "-Rpass-analysis=kernel-resource-usage", "-c", unit + ".hip", "-o", "/dev/null"] + FLAGS.get(unit, []) + sys.argv[2:]
    err = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", dem).replace("void lsr::", "")}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
    for r in rows:
        print(f"{r['name'][:60]:60s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")


if __name__ == "__main__":
    main()
