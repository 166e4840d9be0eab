cd $GRAFT_REPO_ROOT
for b in 1 2 4 6 8; do
  PYMGRIT_AMD_PLAN_BLOCKS=$b timeout -k 10 200 python bench.py --workload advection --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        b=json.loads(ln); print('advection blocks',$b,'ms',round(b['ms_per_step'],3))
"
done
timeout -k 10 200 python bench.py --workload advection --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        b=json.loads(ln); print('advection default ms',round(b['ms_per_step'],3))
"
