#!/bin/bash
# On the GPU box: one rocprofv3 --pmc pass (counters of ONE pass fit the hardware's slots: keep to 2-4) around a python command;
# per kernel the mean of every counter, from the rocpd database, to stdout and gpurun_out/<name>_pmc.txt.
#   bash tools/pmc_pass.sh <name> "SQ_BUSY_CYCLES SQ_WAVES" bench.py --workload heat2d --steps 1 --warmup 0
name=$1; shift
ctrs=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
script=$1; shift
( cd /tmp && timeout -k 10 400 rocprofv3 --pmc $ctrs -d "$out/pmc_$name" -- python3 "$R/$script" "$@" > "$out/${name}_pmc.log" 2>&1 )
python3 - "$out/pmc_$name" "$out/${name}_pmc.txt" <<'PY'
import collections, glob, os, sqlite3, sys
dbs = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*.db")), key=os.path.getmtime)
if not dbs:
    print("no rocpd database under", sys.argv[1]); sys.exit(1)
con = sqlite3.connect(dbs[-1])
rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in rows:
    acc[k][c].append(v)
lines = []
for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    nm = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(", 1024>", ">")[:60]
    lines.append(f"{nm:60s} n={len(next(iter(cs.values()))):5d} " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())))
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
PY
rm -rf "$out/pmc_$name"
