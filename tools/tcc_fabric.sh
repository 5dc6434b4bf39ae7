#!/bin/bash
# tools/tcc_fabric.sh -- what the L2 <-> fabric interface says about the main kernels: request latencies and stall cycles (TCC counters; each
# set its own `rocprofv3 --pmc` run of `bench.py --steps 4`, no other trace domain).  Output: gpurun_out/tcc/<workload>/<set>/ ;
# summarised by tools/tcc_fabric.py
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf $ROOT/gpurun_out/tcc
run() {
    tag=$1; shift
    i=0
    for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" \
               "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
               "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
        out=$ROOT/gpurun_out/tcc/$tag/set$i
        mkdir -p $out
        rocprofv3 --pmc $set --output-format csv -d $out -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
        i=$((i + 1))
    done
    echo "$tag done"
}
run cfg2_fft4096 --no-other-configs
run cfg3_fft1m --workload fft1m
run fft512k --workload fft --n 524288 --radix 2
run fft4m --workload fft --n 4194304 --radix 2
run cfg4_iir --workload iir
run cfg4_iir64 --workload iir64
run fft32768 --workload fft --n 32768 --radix 2
