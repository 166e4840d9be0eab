cd $GRAFT_REPO_ROOT
PYMGRIT_AMD_PLAN_BLOCKS=4 timeout -k 10 600 python -m pytest tests/test_hip_heat2d.py -m gpu -x -q 2>&1 | tail -3
for b in 8 16; do
  PYMGRIT_AMD_PLAN_BLOCKS=$b timeout -k 10 300 python bench.py --workload heat2d --steps 3 --warmup 3 2>gpurun_out/h2d_err_$b.log | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        b=json.loads(ln); print('heat2d graph blocks',$b,'ms',round(b['ms_per_step'],2), 'frac', round(b['roofline']['frac'],3))
"
  tail -3 gpurun_out/h2d_err_$b.log | grep -v amdgpu
done
