#!/bin/bash
# Round 3's profile set, run on the GPU box:  tools/profile_r03.sh   (each tag: bench + rocprofv3 kernel trace + two PMC passes,
# tools/profile.sh; raw output under gpurun_out/prof_<tag>/, condensed by tools/summarize_profile.py <tag> into profiles/)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
tools/profile.sh r03_fft4096 --no-other-configs
tools/profile.sh r03_fft1m --workload fft1m
tools/profile.sh r03_iir --workload iir
tools/profile.sh r03_iir64 --workload iir64
tools/profile.sh r03_fft8192_f64 --workload fft --n 8192 --radix 2 --precision f64
tools/profile.sh r03_fft16384_f64 --workload fft --n 16384 --radix 2 --precision f64
tools/profile.sh r03_fft65536_f64 --workload fft --n 65536 --radix 2 --precision f64
tools/profile.sh r03_fft2m --workload fft --n 2097152 --radix 2
tools/profile.sh r03_fft4m --workload fft --n 4194304 --radix 2
# second part of round 3: the persistent two-pass launch (sdsp_fft2p_fused) and the f64 N = 2^15 two-pass form
tools/profile.sh r03_fft512k --workload fft --n 524288 --radix 2
tools/profile.sh r03_fft32768_f64 --workload fft --n 32768 --radix 2 --precision f64
tools/profile.sh r03_fft1m_f64 --workload fft --n 1048576 --radix 2 --precision f64
