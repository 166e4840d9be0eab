cd $GRAFT_REPO_ROOT
for b in 3 4 5 6 7 8; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --plan-blocks $b --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        b=json.loads(ln); print('blocks',$b,'ms',round(b['ms_per_step'],3),'chain us/step',round(b['config']['chain_us_per_step'],3))
"
done
