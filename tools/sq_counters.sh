#!/bin/bash
# tools/sq_counters.sh -- a few SQ counters of the two tuned N = 4096 kernels (own --pmc runs), to compare them
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
    tag=$1; pass=$2; ctrs=$3; shift 3
    out=$ROOT/gpurun_out/sq/$tag/$pass
    mkdir -p $out
    rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
}
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
B="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
run r4 a "$A"
run r4 b "$B"
run r2 a "$A" --workload fft --n 4096 --radix 2
run r2 b "$B" --workload fft --n 4096 --radix 2
run big8192 a "$A" --workload fft --n 8192 --radix 2
run iir a "$A" --workload iir
echo done
