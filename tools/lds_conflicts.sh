#!/bin/bash
# tools/lds_conflicts.sh -- LDS bank-conflict counters of the main kernels (its own --pmc run, no other trace domain).
# Output: gpurun_out/lds/<tag>/ ; summarised by tools/lds_conflicts.py
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
    tag=$1; shift
    out=$ROOT/gpurun_out/lds/$tag
    mkdir -p $out
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
    echo "$tag done"
}
if [ "$1" = "r03" ]; then # round 3's new kernels (old output of gpurun_out/lds/ is replaced)
    rm -rf $ROOT/gpurun_out/lds
    run iir_landing --workload iir
    run iir_supertile --workload iir --variant 3
    run iir64 --workload iir64
    run fft1m --workload fft1m
    run fft8192_f64 --workload fft --n 8192 --radix 2 --precision f64
    run fft16384_f64 --workload fft --n 16384 --radix 2 --precision f64
    run fft65536_f64 --workload fft --n 65536 --radix 2 --precision f64
    run fft1m_f64 --workload fft --n 1048576 --radix 2 --precision f64
    run fft2m --workload fft --n 2097152 --radix 2
    run fft4m --workload fft --n 4194304 --radix 2
    run fft512k --workload fft --n 524288 --radix 2
    run fft32768_f64 --workload fft --n 32768 --radix 2 --precision f64
    exit 0
fi
run fft4096 --no-other-configs
run fft4096_r2 --workload fft --n 4096 --radix 2
run fft8192 --workload fft --n 8192 --radix 2
run fft8192_mix --workload fft --n 8192 --radix 0
run fft16384 --workload fft --n 16384 --radix 2
run fft1024 --workload fft --n 1024 --radix 2
run fft1m --workload fft1m
run iir --workload iir
run fir32 --workload fir --taps 32
run fft65536 --workload fft --n 65536 --radix 4
