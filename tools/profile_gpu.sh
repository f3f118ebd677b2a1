#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace stats + PMC passes for the k2u batch kernel.
# usage: tools/profile_gpu.sh <workload> <tag> [steps]
# Counters are collected in their own runs (one --pmc set per run), never together with tracing.
set -u
W=${1:-s-eighth}; TAG=${2:-r01}; STEPS=${3:-5}; EXTRA=${4:-}
OUT=gpurun_out/prof_${TAG}_${W}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline $EXTRA"
run_pmc() { # name counters...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc $name failed rc=$?"
}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o trace -- $BENCH > "$OUT/stats.json" 2> "$OUT/stats.err" || echo "stats failed rc=$?"
run_pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run_pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
run_pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
run_pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pmc tcc TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
find "$OUT" -name "*.csv" | head -40
