#!/bin/bash
# Runs ON the GPU box: HBM read requests (128-byte lines) per query for a given query mix.
# usage: tools/pmc_lines.sh <workload> <tag> <extra bench args...>
set -u
W=$1; TAG=$2; shift 2
OUT=gpurun_out/lines_${TAG}_${W}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/pmc" -o pmc -- \
  python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-validate "$@" > "$OUT/bench.json" 2> "$OUT/err.txt" || echo "pmc failed rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    if "k2u_" not in r["Kernel_Name"] or "extract" in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
q = b["config"]["queries_per_gpu_per_step"]
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k[:40]:40s} {c:24s} launches={n} per_launch={v/n:.4g} per_query={v/n/q:.3f}")
print("value", b["value"], "hit", b["hit_fraction"])
PY
