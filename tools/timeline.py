#!/usr/bin/env python3
"""Kernel timeline of the LAST cycles of a rocprofv3 --kernel-trace run: start (relative, ms), duration, stream, name.
usage: tools/timeline.py <dir with the rocpd *.db (or *kernel_trace.csv)> [n_last_kernels]"""
import csv
import glob
import os
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:44]


def rows_of(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if dbs:
        con = sqlite3.connect(dbs[0])
        return [(int(s), int(e), str(st), n) for s, e, st, n in
                con.execute("select start, end, stream_id, name from kernels order by start")]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"])
                  for r in csv.DictReader(open(f)))


def main(d, n_last=140):
    rows = rows_of(d)[-n_last:]
    t0 = rows[0][0]
    for s, e, st, name in rows:
        print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f}  +{(e - s) / 1e6:7.3f} ms  s{st:>3}  {short(name)}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140)
