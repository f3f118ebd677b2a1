#!/bin/bash
# Runs ON the GPU box: instruction counts of the k2u kernel for kernel variants (name=lib.so ...), one rocprofv3 --pmc run each.
# usage: [KERNEL=k2u_reads NQ=120000000] tools/pmc_variants.sh "<ab_kernels args>" name=path.so ...   (counts are per 64 of NQ queries)
set -u
export TMPDIR=/tmp
ARGS=$1; shift
for v in "$@"; do
  n=${v%%=*}
  OUT=/tmp/pmcv_$n
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -o pmc -- \
     python3 tools/ab_kernels.py --steps 3 --reps 1 --no-compare $ARGS $v > $OUT/out.txt 2> $OUT/err.txt || echo "$n failed"
  python3 - $OUT $n ${KERNEL:-k2u_wave} ${NQ:-125000000} <<'PY'
import csv, glob, sys
out, name, kern, nq = sys.argv[1:5]
agg = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]: continue
        a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
q = float(nq)
print(name, " ".join(f"{c.replace('SQ_','')}={v/n/q*64:.0f}" for c, (n, v) in sorted(agg.items())), flush=True)
PY
done
