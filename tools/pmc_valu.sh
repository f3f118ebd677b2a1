#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_r01d
mkdir -p $OUT
B="python3 bench.py --workload s-human --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_a -o pmc -- $B > $OUT/a.json 2> $OUT/a.err
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_b -o pmc -- $B > $OUT/b.json 2> $OUT/b.err
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_c -o pmc -- $B > $OUT/c.json 2> $OUT/c.err
