#!/bin/bash
# VGPRs / scratch / LDS of every kernel in the fused-solver objects (read from the code-object notes).
set -e
B=/opt/rocm/lib/llvm/bin
D=$(mktemp -d)
for o in "$@"; do
  $B/llvm-objcopy --dump-section .hip_fatbin=$D/fat.bin "$o"
  $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$D/fat.bin --output=$D/k.co
  echo "== $o"
  $B/llvm-readelf --notes $D/k.co | grep -E "^\s+\.name:|private_segment_fixed_size|\.vgpr_count|\.vgpr_spill_count|group_segment_fixed_size" \
    | awk '/group_segment/{l=$2} /\.name:/{n=$2} /private_segment/{p=$2} /vgpr_count/{v=$2} /vgpr_spill/{printf "%-60.60s vgpr %4s spill %4s scratch %5s lds %6s\n", n, v, $2, p, l}'
done
rm -rf $D
