#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv of every pass under a directory)
into per-kernel means per launch.  usage: tools/pmc_summary.py gpurun_out/pmc_<tag> > profiles/<name>.md"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:60]


def _kernel_source_hash():
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("render_forward.hip", "lsr_blend.h"):
        with open(os.path.join(root, "latentsplat_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def write_traffic_json(root, acc, path, kernel="lsr::k_render_fwd<4"):
    """profiles/traffic_render_forward.json: HBM bytes per launch of the dominant kernel, read by
    bench.py for roofline.traffic."""
    import json
    # (round 5: the kernel has a RECORD instance for forwards that a backward follows — the headline figure is the plain one)
    for k, counters in sorted(acc.items(), key=lambda kv: ("true>" in kv[0], kv[0])):
        if not k.startswith(kernel):
            continue
        c = {n: v[0] / max(v[1], 1) for n, v in counters.items()}
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        fetch, write = c["FETCH_SIZE"], c["WRITE_SIZE"]
        json.dump({
            "kernel": "k_render_fwd", "instance": k,
            "source": f"{root} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, mean per launch)",
            "fetch_size_kib": fetch, "write_size_kib": write,
            # calibrated in round 4 (tools/microbench/gather_fetch.hip, profiles/r04_fetch_calibration.md): a 64-byte
            # record gather is counted EXACTLY (one 64-byte request per record: 564.6 MB counted for 570.4 MB issued, the
            # difference being the streamed 4-byte index reads, which travel as 128-byte requests tallied at 64 — the only
            # pattern the guide's factor 2 applies to); WRITE_SIZE is exact (2 GiB written, 2 097 152 KiB counted)
            "hbm_bytes_per_launch": (fetch + write) * 1024,
            "hbm_bytes_per_launch_guide_2x": (2 * fetch + write) * 1024,
            "source_commit": os.environ.get("LSR_PROFILE_COMMIT"),
            # the build these counters belong to: bench.py reports the traffic only while the kernel source still hashes to this
            "kernel_source_sha16": _kernel_source_hash(),
            "valu_insts_per_launch": c.get("SQ_INSTS_VALU"),
            "valu_trans_insts_per_launch": c.get("SQ_INSTS_VALU_TRANS_F32"),
            "valu_active_quad_cycles_per_launch": c.get("SQ_ACTIVE_INST_VALU"),
            "note": "FETCH_SIZE + WRITE_SIZE as counted: this kernel's reads are 64-byte record gathers, which the counter "
                    "tallies exactly (calibration above); the guide's 2 x FETCH_SIZE applies to wide streaming reads only "
                    "and is kept as hbm_bytes_per_launch_guide_2x",
        }, open(path, "w"), indent=1)
        return


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if "lsr::" not in k:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    if len(sys.argv) > 3 and sys.argv[2] == "--traffic-json":
        write_traffic_json(root, acc, sys.argv[3])
    print(f"# PMC counters per launch (mean over launches), {root}\n")
    print("FETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE under-counts wide")
    print("coalesced reads by 2x (MI355X_MICROARCH.md §HBM) — `hbm_read_corrected` doubles it.\n")
    for k in sorted(acc):
        print(f"## `{k}`\n")
        print("| counter | mean per launch |")
        print("|---|---:|")
        c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
        for n in sorted(c):
            print(f"| {n} | {c[n]:.4g} |")
        if "FETCH_SIZE" in c:
            print(f"| hbm_read_corrected_bytes (2 x FETCH_SIZE x 1024) | {2 * c['FETCH_SIZE'] * 1024:.4g} |")
        if "WRITE_SIZE" in c:
            print(f"| hbm_write_bytes (WRITE_SIZE x 1024) | {c['WRITE_SIZE'] * 1024:.4g} |")
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
            print(f"| VALU insts per wave | {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.4g} |")
        print()


if __name__ == "__main__":
    main()
