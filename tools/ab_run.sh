#!/bin/bash
# Runs ON the GPU box: same-box A/B of build/ab/<name>.so variants through bench.py (MZ_LIB override).
# usage: tools/ab_run.sh "<bench args>" name1 name2 ...   (each variant is run twice, interleaved)
set -u
ARGS=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for n in "$@"; do
    MZ_LIB=$PWD/build/ab/$n.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-validate $ARGS > gpurun_out/ab/$n.$rep.json 2> gpurun_out/ab/$n.$rep.err || echo "$n failed"
    python3 - $n $rep <<'PY'
import json, sys
n, rep = sys.argv[1:3]
try:
    b = json.loads(open(f"gpurun_out/ab/{n}.{rep}.json").read().strip().splitlines()[-1])
    print(f"{n:12s} rep{rep} value={b['value']:.1f} ms={b['ms_per_step']:.3f} hit={b['hit_fraction']}")
except Exception as e:
    print(n, rep, "ERR", e)
PY
  done
done
