# scratch experiment: chain of a two-group state inside one workgroup (MGRIT_HIP_CHAIN_LOCAL_G) on config 5
for g in 0 2; do
    MGRIT_HIP_CHAIN_LOCAL_G=$g python bench.py --workload advection --steps 10 --warmup 3 > gpurun_out/adv_$g.json 2>gpurun_out/adv_$g.err
    python - <<PY
import json
try:
    b=[json.loads(l) for l in open("gpurun_out/adv_$g.json") if l.startswith("{")][0]
    print("local_g $g", round(b["ms_per_step"],4), b["roofline"]["frac"], {k: round(v["ms_per_cycle"],3) for k,v in list(b["sweeps"].items())[:8]})
except Exception as e: print("local_g $g ERR", e)
PY
done
