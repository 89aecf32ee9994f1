#!/bin/bash
# registers / scratch / LDS / code size of the kernels of a built library whose name matches a pattern
#   tools/kres.sh linesearch [lib]
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d); lib=${2:-upright_amd/libupright_mi.so}
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin $lib && $L/clang-offload-bundler --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
$L/llvm-readelf --notes $T/dev.co | awk -v pat="$1" '/\.name:/ {name=$2} /\.private_segment_fixed_size:/ {ps=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2} /\.agpr_count:/ {ag=$2} /\.vgpr_spill_count:/ {sp=$2; if (name ~ pat) print name, "vgpr", vg, "agpr", ag, "spill", sp, "scratch", ps}'
$L/llvm-readelf -s $T/dev.co | awk -v pat="$1" '$8 ~ pat && $8 !~ /\.(kd|num_vgpr|num_agpr|numbered_sgpr|private_seg_size|uses_vcc|uses_flat_scratch|has_dyn_sized_stack|has_recursion|has_indirect_call)$/ {print $8, "code bytes", $3}'
cp $T/dev.co /tmp/isa/dev.co 2>/dev/null; rm -rf $T
