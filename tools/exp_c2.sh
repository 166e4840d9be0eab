# scratch experiment: automatic chunk length of the level-0 whole-level passes
for cfg in "1024 4097" "4096 16385" "16384 65537"; do
    set -- $cfg
    st=200; [ $1 = 16384 ] && st=30
    python bench.py --nx $1 --nt $2 --steps $st --warmup 10 --no-cpu-baseline > gpurun_out/ch_$1_auto.json 2>gpurun_out/ch_$1_auto.err
    python - <<PY
import json
try:
    b=[json.loads(l) for l in open("gpurun_out/ch_$1_auto.json") if l.startswith("{")][0]
    print("nx $1 auto", round(b["ms_per_step"],4), {k: round(v["ms_per_cycle"],4) for k,v in b["sweeps"].items()})
except Exception as e: print("nx $1 auto ERR", e)
PY
done
