#!/bin/bash
# Register / scratch / STATIC LDS use of every kernel in one csrc file, from the compiler's .amdhsa metadata:
#   tools/kernel_regs.sh fft_big.hip [extra hipcc flags]
# (vgpr = .amdhsa_next_free_vgpr: architectural VGPRs incl. the AGPR block above accum_off; "lds" is the static segment only -- the
# kernels here take dynamic LDS, sized by their launchers: DESIGN.md section 5)
src=$1; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Isimpledsp_amd/csrc -fno-slp-vectorize "$@" \
    --cuda-device-only -S -o $tmp/k.s simpledsp_amd/csrc/$src 2>/dev/null
awk '$1==".amdhsa_kernel"{name=$2} $1==".amdhsa_next_free_vgpr"{v=$2} $1==".amdhsa_accum_offset"{a=$2} $1==".amdhsa_private_segment_fixed_size"{p=$2} $1==".amdhsa_group_segment_fixed_size"{l=$2} $1==".end_amdhsa_kernel"{printf "%s vgpr %s accum_off %s scratch %s lds %s\n", name, v, a, p, l}' $tmp/k.s | c++filt | sed 's/sdsp_hip::(anonymous namespace):://; s/(HIP_vector_type.*) / /'
rm -rf $tmp
