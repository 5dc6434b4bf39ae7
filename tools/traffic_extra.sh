#!/bin/bash
# tools/traffic_extra.sh -- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the non-BASELINE kernels.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
    tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        out=$ROOT/gpurun_out/traffic/$tag/$c
        mkdir -p $out
        rocprofv3 --pmc $c --output-format csv -d $out -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
    done
    echo "$tag done"
}
run fft8192 --workload fft --n 8192 --radix 2
run fft16384 --workload fft --n 16384 --radix 2
run fft1024 --workload fft --n 1024 --radix 4
run fft65536 --workload fft --n 65536 --radix 4
run fir32 --workload fir --taps 32
run iir_il --workload iir_il
