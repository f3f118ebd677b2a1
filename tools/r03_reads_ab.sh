#!/bin/bash
# Runs ON the GPU box: reads-mode bench for library variants x verification forms, same box, interleaved twice
# usage: [LIBS="tree sf8"] [VERIFY="group each"] tools/r03_reads_ab.sh [workload ...]
for w in "${@:-s-human s-txome}"; do for rep in 1 2; do for n in ${LIBS:-tree}; do for v in ${VERIFY:-group each}; do
LIB=$PWD/build/ab/$n.so; [ "$n" = tree ] && LIB=$PWD/mazu_amd/libmazu_amd.so
MZ_LIB=$LIB MZ_READS_VERIFY=$v timeout -k 10 200 python3 bench.py --mode reads --no-cpu-baseline --workload $w --steps 10 ${BENCHX:-} 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $n $v', b['value'], b['ms_per_step'], b['hit_fraction'], flush=True)" || exit 1
done; done; done; done
