#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats around a python command, the kernel table (from the rocpd database) into
# gpurun_out/<name>_kernel_stats.csv and its first lines to stdout; the database itself is dropped.
#   bash tools/quick_prof.sh <name> bench.py --steps 10 --warmup 3 ...
name=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
script=$1; shift
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/qp_$name" -- python3 "$R/$script" "$@" > "$out/${name}_prof.log" 2>&1 )
python3 - "$out/qp_$name" "$out/${name}_kernel_stats.csv" <<'PY'
import glob, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tools"))
from summarize_profiles import stats_from_db
dbs = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*.db")), key=os.path.getmtime)
if not dbs:
    print("no rocpd database under", sys.argv[1]); sys.exit(1)
stats_from_db(dbs[-1], sys.argv[2])
for ln in open(sys.argv[2]).read().splitlines()[:16]:
    print(ln[:230])
PY
rm -rf "$out/qp_$name"
