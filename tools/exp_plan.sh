cd $GRAFT_REPO_ROOT
for c in 1 2; do for b in 5 6 7; do
  PYMGRIT_AMD_FUSE_CHUNK_COARSE=$c timeout -k 10 120 python bench.py --no-cpu-baseline --plan-blocks $b --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        b=json.loads(ln); print('coarse chunk',$c,'blocks',$b,'ms',round(b['ms_per_step'],3),'chain us/step',round(b['config']['chain_us_per_step'],3), 'ecL1',round(b['sweeps']['ec_relax L1']['ms_per_cycle'],3))
"
done; done
