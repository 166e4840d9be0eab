#!/bin/bash
# SQ counters of the sweep kernels (instruction mix, busy / wait cycles): two passes, program-order cycle
set -o pipefail
out=$PWD/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp; B="$PWD/bench.py"; cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU -d "$out/sq1" -- python3 "$B" --steps 3 --warmup 1 --no-cpu-baseline --no-ramp --plan-blocks 1 > "$out/sq1.log" 2>&1 || { tail -5 "$out/sq1.log"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d "$out/sq2" -- python3 "$B" --steps 3 --warmup 1 --no-cpu-baseline --no-ramp --plan-blocks 1 > "$out/sq2.log" 2>&1 || { tail -5 "$out/sq2.log"; exit 2; }
ls "$out/sq1" "$out/sq2"
