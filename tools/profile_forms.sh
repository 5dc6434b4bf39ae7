#!/bin/bash
# tools/profile_forms.sh -- rocprofv3 kernel-trace stats of the registers-resident kernel's fused convolution and real-input
# forms (run on the GPU box; raw output under gpurun_out/prof_forms/, condensed by hand into profiles/r02_fft_big_forms.md)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_forms
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conv8k -- python3 $ROOT/tools/conv_probe.py 8192 131072 2 > $OUT/conv8k.log 2> $OUT/conv8k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conv16k -- python3 $ROOT/tools/conv_probe.py 16384 65536 2 > $OUT/conv16k.log 2> $OUT/conv16k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rfft -- python3 $ROOT/tools/bench_rfft.py 8192,16384,32768,65536 > $OUT/rfft.log 2> $OUT/rfft.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rfft_fetch -- python3 $ROOT/tools/bench_rfft.py 16384 > $OUT/rfft_fetch.log 2> $OUT/rfft_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rfft_write -- python3 $ROOT/tools/bench_rfft.py 16384 > $OUT/rfft_write.log 2> $OUT/rfft_write.err
find $OUT -name "*kernel_stats.csv" | head
