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
sys.path.insert(0, ROOT)


def stamp():
    """the build the counters were collected on (bench.build_stamp: sources + library): bench.py prices physical bytes only from a
    traffic file whose stamp is that of the running build"""
    from bench import build_stamp
    return build_stamp()


def short(name):
    """kernel name without namespace, return type and argument list; the sweeps' default workgroup-size argument (", 1024>", round 5)
    is dropped, so that a kernel keeps the name earlier rounds' files know it by (the single-wave instances keep their ", 64>")"""
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(", 1024>", ">")


def newest(paths):
    """gpurun merges every call's output into the same directories: keep the most recent database only"""
    return sorted(paths, key=os.path.getmtime)[-1:]


def stats_from_db(db, dest):
    """rocprofv3 of ROCm 7.2 writes a rocpd SQLite database instead of CSV files: rebuild the --stats kernel table
    (same columns as <pid>_kernel_stats.csv) from its `kernels` view."""
    import sqlite3
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(dest, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, int(tot), f"{avg:.3f}", f"{100.0 * tot / total:.6f}", int(mn), int(mx)])


def counters_from_db(db, counter, by_workgroup=False):
    import sqlite3
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, value, workgroup_size_x, grid_size from counters_collection where counter_name = ?", (counter,)).fetchall()
    # by_workgroup: one template instance serves several levels (the general whole-level passes of config 5): its launches differ
    # in the workgroup size (64 threads per group of 1024 values), which keeps the levels apart; "grid": by the launch's total size
    # (wide states: the batches of a sweep against the small launches around them)
    if by_workgroup == "grid":
        return [(f"{short(n)}@grid{g}", v) for n, v, wg, g in rows]
    return [(f"{short(n)}@{wg}" if by_workgroup else n, v) for n, v, wg, g in rows]


def traffic_summary(tag, suffix, by_workgroup):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for db in newest(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{which}{suffix}", "*", "*.db"))):
            for name, value in counters_from_db(db, counter, by_workgroup):
                per[name if by_workgroup else short(name)][counter].append(float(value))
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
    return summary


def main(tag):
    out = os.path.join(ROOT, "profiles")
    for sub in ("stats", "stats_graph", "stats_config2", "stats_heat2d", "stats_advection", "stats_rank3of8", "stats_wide"):
        dest = os.path.join(out, f"{tag}_kernel_{sub}.csv")
        stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{sub}", "*", "*kernel_stats.csv"))
        dbs = newest(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{sub}", "*", "*.db")))
        if stats:
            shutil.copy(stats[0], dest)
        elif dbs:
            stats_from_db(dbs[0], dest)
    for log in glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_bench*.log")):
        lines = [ln for ln in open(log) if ln.startswith("{")]
        if lines:
            with open(os.path.join(out, os.path.basename(log).replace(".log", ".json")), "w") as f:
                f.writelines(lines)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for db in newest(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_{which}", "*", "*.db"))):
            for name, value in counters_from_db(db, counter):
                per[short(name)][counter].append(float(value))
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
        json.dump({"build": stamp(),
                   "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around "
                           "`python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ramp --plan-blocks 1` (tools/profile_round.sh); KiB units; read side doubled "
                           "(gfx950 FETCH_SIZE tallies 128-B requests at 64 B). Every kernel name of the headline cycle belongs to one (sweep, level): cfas/ecfr = level 0, the <.., true, ..> kernels and fas_fused1 = level 1, chain2 = level 2; relax_kernel<1, 1, false, 0|1> = the stand-alone level-0 F-/C-relax launches of the fcf_relax_level0 figure.",
                   "kernels": summary}, f, indent=1)
    adv = traffic_summary(tag, "_advection", True)
    if adv:
        with open(os.path.join(out, f"{tag}_traffic_advection.json"), "w") as f:
            json.dump({"build": stamp(),
                       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python3 bench.py --workload advection "
                               "--steps 3 --warmup 2` (tools/profile_round.sh); KiB units; read side doubled (gfx950 FETCH_SIZE tallies 128-B "
                               "requests at 64 B). Keys: kernel@workgroup size -- 512 threads = level 0 (8192 values), 256 = level 1, "
                               "128 = levels 2 and 3 (2048 values).", "kernels": adv}, f, indent=1)
    wide = {k: v for k, v in traffic_summary(tag, "_wide", "grid").items() if k.startswith("wide_")}
    if wide:
        with open(os.path.join(out, f"{tag}_traffic_wide.json"), "w") as f:
            json.dump({"build": stamp(),
                       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python3 tools/wide_bench.py` (Heat1D nx=32770, "
                               "nt=4097: wide states, three launches per Phi); KiB units; read side doubled (gfx950 FETCH_SIZE tallies 128-B "
                               "requests at 64 B). Keys: kernel@grid<threads of the launch>. The level-0 F-relaxation of 3072 Phi = three batches of 1024 rows, each "
                               "wide_local_kernel<2>@grid2097152 + wide_carry_kernel@grid65536 + wide_finish_kernel@grid2097152.",
                       "kernels": wide}, f, indent=1)
    print("wrote", os.listdir(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
