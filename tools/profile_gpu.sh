#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace stats + PMC passes for the k2u kernels of one bench.py command.
# usage: tools/profile_gpu.sh <workload> <tag> [steps] ["extra bench.py args"]
# Counters are collected in their own runs (one --pmc set per run), never together with tracing.  Raw profiler output stays in
# /tmp on the box; gpurun_out/prof_<tag>_<workload>/ gets summary.txt, kernel_stats.csv and the bench lines printed under the profiler.
set -u
W=${1:-s-eighth}; TAG=${2:-r02}; STEPS=${3:-5}; EXTRA=${4:-}
RAW=/tmp/prof_${TAG}_${W}; OUT=gpurun_out/prof_${TAG}_${W}
rm -rf "$RAW"; mkdir -p "$RAW" "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline $EXTRA"
run_pmc() { # name counters...
  local name=$1; shift
  echo "pass pmc_$name" >&2
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d "$RAW/pmc_$name" -o pmc -- $BENCH > "$RAW/pmc_$name.json" 2> "$RAW/pmc_$name.err" || echo "pmc $name failed rc=$?"
}
echo "pass stats" >&2
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/stats" -o trace -- $BENCH > "$RAW/stats.json" 2> "$RAW/stats.err" || echo "stats failed rc=$?"
run_pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run_pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
run_pmc fetch FETCH_SIZE TCC_REQ_sum GRBM_GUI_ACTIVE
run_pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pmc tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run_pmc lat TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum
{
  echo "command: bash tools/profile_gpu.sh $W $TAG $STEPS $EXTRA   (separate rocprofv3 passes)"
  echo "bench line under --kernel-trace: $(tail -1 "$RAW/stats.json" | cut -c1-400)"
  python3 tools/summarize_prof.py "$RAW" k2u
} > "$OUT/summary.txt"
KS=$(find "$RAW/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats.csv"
tail -1 "$RAW/stats.json" > "$OUT/bench_line_under_profiler.json"
cat "$OUT/summary.txt"
