#!/bin/bash
# Runs ON the GPU box: HBM read requests per query of the k2u kernel for A/B variants (build/ab/<name>.so).
# usage: tools/pmc_lines_ab.sh "<bench args>" name1 name2 ...
set -u
export TMPDIR=/tmp
ARGS=$1; shift
for n in "$@"; do
  OUT=gpurun_out/linesab_$n
  mkdir -p $OUT
  export MZ_LIB=$PWD/build/ab/$n.so
  timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc -o pmc -- \
     python3 bench.py --workload s-human --steps 2 --warmup 1 --no-cpu-baseline --no-validate $ARGS > $OUT/bench.json 2> $OUT/err.txt || echo "$n failed"
  python3 - $OUT $n <<'PY'
import csv, glob, sys
out, name = sys.argv[1:3]
agg = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2u_wave" not in r["Kernel_Name"]: continue
        a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
q = 125e6
print(name, " ".join(f"{c}={v/n/q:.3f}" for c, (n, v) in sorted(agg.items())))
PY
done
