#!/bin/bash
# Runs ON the GPU box: dynamic instruction counts of the reads kernel (per 64 windows) for A/B variants (build/ab/<name>.so; "tree" =
# the in-tree library).  Extra environment for the bench through ENVX="A=1 B=2".
# usage: tools/pmc_reads_insts.sh <workload> name1 name2 ...
set -u
export TMPDIR=/tmp
W=$1; shift
for kv in ${ENVX:-}; do export "$kv"; done
for n in "$@"; do
  OUT=/tmp/insts_reads_$n; rm -rf $OUT; mkdir -p $OUT
  LIB=$PWD/build/ab/$n.so; [ "$n" = tree ] && LIB=$PWD/mazu_amd/libmazu_amd.so
  export MZ_LIB=$LIB
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -o pmc -- \
     python3 bench.py --mode reads --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-validate ${BENCHX:-} > $OUT/bench.json 2> $OUT/err.txt || { echo "$n failed"; tail -n 3 $OUT/err.txt; exit 1; }
  python3 - $OUT $n <<'PY'
import csv, glob, json, sys
out, name = sys.argv[1:3]
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
q = b["config"]["windows_per_gpu_per_step"]
agg = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2u_reads" not in r["Kernel_Name"]: continue
        a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
print(name, "per 64 windows:", " ".join(f"{c.replace('SQ_','')}={v/n/q*64:.0f}" for c, (n, v) in sorted(agg.items())), "| ms", b["ms_per_step"], flush=True)
PY
done
