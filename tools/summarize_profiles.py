#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/<tag>_stats, <tag>_fetch, <tag>_write) into the committed summaries
under profiles/: <tag>_kernel_stats.csv (verbatim --stats table), <tag>_traffic.json (per-kernel HBM bytes per launch
from the FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of MI355X_MICROARCH.md: counters are in KiB,
FETCH_SIZE reports 1/2 of a wide coalesced read stream)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def main(tag):
    out = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_stats", "*", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{which}", "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    per[short(r["Kernel_Name"])][counter].append(float(r["Counter_Value"]))
    if not per:
        print("no counter passes found under gpurun_out/: profiles/ left untouched")
        return
    summary = {}
    for k, d in per.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        f_kib = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        w_kib = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        summary[k] = {"launches_fetch_pass": len(d["FETCH_SIZE"]), "launches_write_pass": len(d["WRITE_SIZE"]),
                      "FETCH_SIZE_KiB_avg": f_kib, "WRITE_SIZE_KiB_avg": w_kib,
                      "read_bytes_per_launch_corrected": 2.0 * f_kib * 1024.0, "write_bytes_per_launch": w_kib * 1024.0,
                      "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0}
    with open(os.path.join(out, f"{tag}_traffic.json"), "w") as f:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around "
                           "`python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ramp`; KiB units; read side doubled "
                           "(gfx950 FETCH_SIZE tallies 128-B requests at 64 B). Averages mix levels for kernels that "
                           "are launched on several levels; relax_kernel<1, 1, false, 0> is level-0 F-relax only.",
                   "kernels": summary}, f, indent=1)
    print("wrote", os.listdir(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
