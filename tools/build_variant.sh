#!/bin/bash
# Developer aid: build an A/B variant of the library next to libcfs_hip.so without touching it.
# usage: tools/build_variant.sh <name> "<extra flags for the fused-solver objects>" ["<W2M_FLAGS>" ["<W2S_FLAGS>"]]   -> motionplanning_5d_m_amd/libcfs_<name>.so
set -e
D=$(cd "$(dirname "$0")/../motionplanning_5d_m_amd/csrc" && pwd)
T=$(mktemp -d)
cp "$D"/*.hip "$D"/*.h "$D"/Makefile "$T"/
if [ -n "$FUSED_SRC" ]; then cp "$FUSED_SRC" "$T/cfs_fused.hip"; fi   # A/B against another revision of the fused solver
if [ -n "$DEVICE_H" ]; then cp "$DEVICE_H" "$T/cfs_device.h"; fi
mkdir -p "$T/../include_stub"
sed -i 's#\.\./\.\./include/cfs_hip.h#'"$D"'/../../include/cfs_hip.h#g' "$T"/Makefile "$T"/cfs_device.h
for o in cfs_api cfs_geom cfs_gemm cfs_mesh cfs_chomp cfs_rrt; do cp "$D/$o.o" "$T/" 2>/dev/null && touch "$T/$o.o"; done
make -C "$T" -j8 -s XFLAGS="$2" ${3:+W2M_FLAGS="$3"} ${4:+W2S_FLAGS="$4"} OUTNAME="libcfs_$1.so" OUT="$D/../libcfs_$1.so"
rm -rf "$T"
echo "built $D/../libcfs_$1.so"
