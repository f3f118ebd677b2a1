#!/bin/bash
# Runs ON the GPU box: size classes of the L2 -> fabric read requests and their average latency for the k2u kernel.
# usage: tools/pmc_ea.sh <tag> [bench args...]
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/ea_$TAG
mkdir -p $OUT
B="python3 bench.py --workload s-human --steps 3 --warmup 1 --no-cpu-baseline --no-validate $*"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_a -o pmc -- $B > $OUT/a.json 2> $OUT/a.err || echo "a failed"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -o pmc -- $B > $OUT/b.json 2> $OUT/b.err || echo "b failed"
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for sub in ("pmc_a", "pmc_b"):
    agg = {}
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k2u_wave" not in r["Kernel_Name"]: continue
            a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    print(sub, " ".join(f"{c}={v/n:.4g}" for c, (n, v) in sorted(agg.items())))
PY
