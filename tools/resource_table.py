#!/usr/bin/env python3
"""Kernel resource table from a `hipcc -Rpass-analysis=kernel-resource-usage` log: name, VGPRs, AGPRs, spilled SGPRs / VGPRs,
scratch bytes per lane, occupancy, LDS. Usage: tools/resource_table.py <log> [substring ...] > profiles/rNN_kernel_resources.txt"""
import re
import subprocess
import sys


def main():
    log = open(sys.argv[1]).read()
    want = sys.argv[2:]
    rows = []
    for blk in log.split("Function Name: ")[1:]:
        name = blk.split()[0]
        def f(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        rows.append((name, f("VGPRs"), f("AGPRs"), f("SGPRs Spill"), f("VGPRs Spill"), f(r"ScratchSize \[bytes/lane\]"),
                     f(r"Occupancy \[waves/SIMD\]"), f(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':70s} VGPR AGPR sgprSpill vgprSpill scratchB occ LDS")
    for r, nm in zip(rows, names):
        nm = re.sub(r"^void \(anonymous namespace\)::", "", nm)
        nm = re.sub(r"\(.*$", "", nm)
        if want and not any(w in nm for w in want):
            continue
        print(f"{nm[:70]:70s} {r[1]:4d} {r[2]:4d} {r[3]:9d} {r[4]:9d} {r[5]:8d} {r[6]:3d} {r[7]}")


if __name__ == "__main__":
    main()
