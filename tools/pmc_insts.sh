#!/bin/bash
# Runs ON the GPU box: dynamic instruction counts of the k2u kernel for A/B variants (build/ab/<name>.so).
# usage: tools/pmc_insts.sh name1 name2 ...
set -u
export TMPDIR=/tmp
for n in "$@"; do
  OUT=gpurun_out/insts_$n
  mkdir -p $OUT
  export MZ_LIB=$PWD/build/ab/$n.so
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -o pmc -- \
     python3 bench.py --workload s-human --steps 2 --warmup 1 --no-cpu-baseline --no-validate > $OUT/bench.json 2> $OUT/err.txt || echo "$n failed"
  python3 - $OUT $n <<'PY'
import csv, glob, sys
out, name = sys.argv[1:3]
agg = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2u_wave" not in r["Kernel_Name"]: continue
        a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
q = 125e6
print(name, " ".join(f"{c.replace('SQ_','')}={v/n/q*64:.0f}" for c, (n, v) in sorted(agg.items())))
PY
done
