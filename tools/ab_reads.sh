#!/bin/bash
# Runs ON the GPU box: reads-mode bench for kernel variants (MZ_LIB override), each twice; extra env via ENVX="A=1 B=2"
# usage: tools/ab_reads.sh "<bench args>" name1 name2 ...   (name "tree" = the in-tree library)
set -u
ARGS=$1; shift
for rep in 1 2; do
  for n in "$@"; do
    LIB=$PWD/build/ab/$n.so; [ "$n" = tree ] && LIB=$PWD/mazu_amd/libmazu_amd.so
    MZ_LIB=$LIB timeout -k 10 300 python3 bench.py --mode reads --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n rep$rep', b['value'], b['ms_per_step'], b['hit_fraction'])"
  done
done
