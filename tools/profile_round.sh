#!/bin/bash
# Run on the GPU box (gpurun): the profiles a round commits under profiles/ (tools/summarize_profiles.py <tag> copies them).
#   <tag>_stats          rocprofv3 --kernel-trace --stats, cycle in program order (one launch per sweep and level)
#   <tag>_stats_graph    the same cycle planned as one block and replayed as ONE hipGraph (PYMGRIT_AMD_PLAN_GRAPH=1: opt-in since round 4,
#                        where the replay measured no faster than the launches issued one by one)
#   <tag>_fetch/_write   separate --pmc FETCH_SIZE / WRITE_SIZE passes (program order)
#   <tag>_bench*.log     plain bench lines (default run with the CPU baseline; --all-configs)
# (Round 4: the coarsest-level solves are time-parallel -- DESIGN.md 3.8 --, so no workload's default cycle uses CU-masked streams
# any more and the Heat2D profile is the path the bench times.)
set -o pipefail
tag=${1:-r05}
from=${2:-1}     # first step to run (a rerun after a failed step: bash tools/profile_round.sh r03 7)
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
B="$PWD/bench.py"
cd /tmp
if [ "$from" -le 2 ]; then timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats" -- python3 "$B" --steps 10 --warmup 3 --no-cpu-baseline --no-ramp --plan-blocks 1 > "$out/${tag}_bench_program_order.log" 2>&1 || exit 2; fi
if [ "$from" -le 3 ]; then PYMGRIT_AMD_PLAN_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats_graph" -- python3 "$B" --steps 10 --warmup 3 --no-cpu-baseline --no-ramp > "$out/${tag}_bench_graph.log" 2>&1 || exit 3; fi
if [ "$from" -le 4 ]; then timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_fetch" -- python3 "$B" --steps 4 --warmup 2 --no-cpu-baseline --no-ramp --plan-blocks 1 > "$out/${tag}_fetch.log" 2>&1 || exit 4; fi
if [ "$from" -le 5 ]; then timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$out/${tag}_write" -- python3 "$B" --steps 4 --warmup 2 --no-cpu-baseline --no-ramp --plan-blocks 1 > "$out/${tag}_write.log" 2>&1 || exit 5; fi
# the other BASELINE configurations and one emulated rank of the sharded run: kernel summaries
if [ "$from" -le 6 ]; then timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats_config2" -- python3 "$B" --nx 1024 --nt 4097 --steps 100 --warmup 10 --no-cpu-baseline --no-ramp > "$out/${tag}_bench_config2.log" 2>&1 || exit 6; fi
if [ "$from" -le 7 ]; then timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats_heat2d" -- python3 "$B" --workload heat2d --steps 2 --warmup 1 > "$out/${tag}_bench_heat2d.log" 2>&1 || exit 7; fi
if [ "$from" -le 8 ]; then timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats_advection" -- python3 "$B" --workload advection --steps 10 --warmup 3 > "$out/${tag}_bench_advection.log" 2>&1 || exit 8; fi
if [ "$from" -le 9 ]; then timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_fetch_advection" -- python3 "$B" --workload advection --steps 3 --warmup 2 > "$out/${tag}_fetch_advection.log" 2>&1 || exit 13; fi
if [ "$from" -le 10 ]; then timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$out/${tag}_write_advection" -- python3 "$B" --workload advection --steps 3 --warmup 2 > "$out/${tag}_write_advection.log" 2>&1 || exit 14; fi
if [ "$from" -le 11 ]; then timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats_rank3of8" -- python3 "$B" --emulate-rank 3/8 --steps 20 --warmup 3 > "$out/${tag}_bench_rank3of8.log" 2>&1 || exit 9; fi
if [ "$from" -le 12 ]; then timeout -k 10 300 python3 "$B" --emulate-rank all/8 --steps 20 --warmup 3 > "$out/${tag}_bench_all8.log" 2> "$out/${tag}_bench_all8.err" || exit 10; fi
if [ "$from" -le 13 ]; then timeout -k 10 300 python3 "$B" --emulate-rank all/4 --steps 20 --warmup 3 > "$out/${tag}_bench_all4.log" 2> "$out/${tag}_bench_all4.err" || exit 11; fi
if [ "$from" -le 14 ]; then timeout -k 10 300 python3 "$B" --emulate-rank all/2 --steps 20 --warmup 3 > "$out/${tag}_bench_all2.log" 2> "$out/${tag}_bench_all2.err" || exit 12; fi
if [ "$from" -le 15 ]; then for P in 2 4 8; do timeout -k 10 300 python3 "$B" --emulate-solve $P > "$out/${tag}_bench_solve$P.log" 2> "$out/${tag}_bench_solve$P.err" || exit 15; done; fi
if [ "$from" -le 15 ]; then for P in 2 4 8; do timeout -k 10 200 python3 "$B" --workload advection --emulate-rank all/$P --steps 10 --warmup 3 > "$out/${tag}_bench_advection_all$P.log" 2> "$out/${tag}_bench_advection_all$P.err" || exit 15; done; fi
W="$(dirname "$B")/tools/wide_bench.py"
if [ "$from" -le 16 ]; then timeout -k 10 200 python3 "$W" > "$out/${tag}_bench_wide.log" 2> "$out/${tag}_bench_wide.err" || exit 16; fi
if [ "$from" -le 17 ]; then timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/${tag}_stats_wide" -- python3 "$W" > "$out/${tag}_stats_wide.log" 2>&1 || exit 17; fi
if [ "$from" -le 18 ]; then timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$out/${tag}_fetch_wide" -- python3 "$W" > "$out/${tag}_fetch_wide.log" 2>&1 || exit 18; fi
if [ "$from" -le 19 ]; then timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$out/${tag}_write_wide" -- python3 "$W" > "$out/${tag}_write_wide.log" 2>&1 || exit 19; fi
# the summaries are made HERE, on the GPU box (gpurun copies at most 64 MiB back, the rocprofv3 databases are larger): they go to
# gpurun_out/<tag>_profiles/, from where `cp gpurun_out/<tag>_profiles/* profiles/` takes them; then the databases are dropped
cd "$(dirname "$B")" && python3 tools/summarize_profiles.py "$tag" > "$out/${tag}_summarize.log" 2>&1
# the bench lines that price PHYSICAL bytes come last: they read profiles/<tag>_traffic*.json, which the summary above has just made
# from the counter passes of this very build (made first, the lines would find the file of an older build and say "stale")
cd /tmp
if [ "$from" -le 20 ]; then timeout -k 10 500 python3 "$B" --all-configs > "$out/${tag}_bench_all.log" 2> "$out/${tag}_bench_all.err" || exit 1; fi
if [ "$from" -le 21 ]; then timeout -k 10 200 python3 "$B" --workload advection --steps 10 --warmup 3 > "$out/${tag}_bench_advection.log" 2> "$out/${tag}_bench_advection.err" || exit 21; fi
if [ "$from" -le 22 ]; then timeout -k 10 200 python3 "$B" --workload advection --nx-adv 8001 --steps 10 --warmup 3 > "$out/${tag}_bench_advection_nx8001.log" 2> "$out/${tag}_bench_advection_nx8001.err" || exit 22; fi
cd "$(dirname "$B")" && python3 tools/summarize_profiles.py "$tag" > "$out/${tag}_summarize2.log" 2>&1
# round 5: the FP64 matrix rate this card sustains (register operands / LDS operands), the matrix-pipe counters of the Heat2D
# transforms (one --pmc pass), the phase timeline of the one-launch block solve at config 2 (stamped experiment build, if built:
# python tools/blk_timeline.py build)
if [ "$from" -le 23 ] && [ -x tools/micro/mfma_f64_peak ]; then timeout -k 10 120 tools/micro/mfma_f64_peak > "profiles/${tag}_mfma_f64_peak.txt" 2>&1 || true; fi
if [ "$from" -le 24 ]; then bash tools/pmc_pass.sh "${tag}_heat2d" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" bench.py --workload heat2d --steps 1 --warmup 0 --nt2d 4097 > /dev/null 2>&1; cp "$out/${tag}_heat2d_pmc.txt" "profiles/${tag}_pmc_heat2d.txt" 2>/dev/null || true; fi
if [ "$from" -le 25 ] && [ -f scratch/blk_timeline/libmgrit_hip_stamps.so ]; then timeout -k 10 200 python3 tools/blk_timeline.py run_small > "profiles/${tag}_blk_one_timeline.txt" 2> "$out/${tag}_blk_one_timeline.err" || true; fi
mkdir -p "$out/${tag}_profiles" && cp profiles/${tag}_* "$out/${tag}_profiles/"
find "$out" -name '*_results.db' -delete
find "$out" -name '*kernel_trace.csv' -size +20M -delete
ls -R "$out" | head -50
