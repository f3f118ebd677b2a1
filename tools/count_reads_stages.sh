#!/bin/bash
# Runs ON the GPU box: per-launch work counts of the reads kernel (120 M windows), read from the `skipped` tally of scratch builds
#   for n in 1 2 3 4; do tools/ab_build.sh dbg$n -DMZ_DEBUG_COUNT=$n; done     (1 = bucket entries scanned in S3, 2 = candidates verified
#   in S5 -- sequence records gathered, with grouped verification --, 3 = exact walks after a failed verification, 4 = runs of equal
#   minimizers, i.e. minimizer lookups)
# usage: [VARIANTS="2 3"] [ENVX="A=1"] tools/count_reads_stages.sh [workload ...]
for kv in ${ENVX:-}; do export "$kv"; done
for w in "${@:-s-txome}"; do for v in ${VARIANTS:-1 2 3 4}; do
MZ_LIB=$PWD/build/ab/dbg$v.so timeout -k 10 200 python3 bench.py --mode reads --no-cpu-baseline --workload $w --steps 1 --warmup 0 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w dbg$v ${ENVX:-}', b['skipped'], b['hit_fraction'], flush=True)"
done; done
