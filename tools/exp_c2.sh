python scratch/cfas_abl.py 2>&1 | grep abl
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ramp > gpurun_out/nt.json 2>gpurun_out/nt.err
python - <<PY
import json
b=[json.loads(l) for l in open("gpurun_out/nt.json") if l.startswith("{")][0]
print("planned", round(b["ms_per_step"],4), {k: round(v["ms_per_cycle"],3) for k,v in b["sweeps"].items() if v["ms_per_cycle"]>0.1})
PY
python bench.py --workload advection --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
b=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('adv', b['ms_per_step'], b['roofline']['frac'])"
python bench.py --nx 1024 --nt 4097 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=[json.loads(l) for l in sys.stdin if l.startswith('{')][0]
print('c2', b['ms_per_step'])"
