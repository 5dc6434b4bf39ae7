#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- run on the GPU box: bench + rocprofv3 kernel trace + PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate runs, never combined with other trace domains).
# Raw output lands in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns it into profiles/.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $ROOT/bench.py --steps 100 --warmup 10 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench_traced.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write done"
find $OUT -type f | head -40
