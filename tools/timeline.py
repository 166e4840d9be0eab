#!/usr/bin/env python3
"""Kernel timeline of the LAST cycle of a rocprofv3 --kernel-trace run: start (relative, ms), duration, queue, name.
usage: tools/timeline.py <dir with *kernel_trace.csv> [n_last_kernels]"""
import csv
import glob
import os
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]


def main(d, n_last=140):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n_last:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e6:9.3f} +{(e - s) / 1e6:7.3f} ms  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', '?'):>7}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140)
