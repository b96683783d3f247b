#!/bin/bash
# Builds librtr_hip.so of the WORKING TREE with extra compiler flags next to the normal one, for same-box A/B runs (profiles/ab_lib.sh):
#   bash profiles/build_flags_variant.sh "-DRTR_TRI_PAIRS=1" pairs     ->  realtimeraytracer_amd/librtr_hip_pairs.so
set -e
FLAGS=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/rtr_variant_XXXX)
mkdir -p "$TMP/realtimeraytracer_amd"
cp -r "$ROOT/include" "$TMP/include"
cp -r "$ROOT/realtimeraytracer_amd/csrc" "$TMP/realtimeraytracer_amd/csrc"
rm -rf "$TMP/realtimeraytracer_amd/csrc/build"
make -C "$TMP/realtimeraytracer_amd/csrc" -j8 ../librtr_hip.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value $FLAGS" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
cp "$TMP/realtimeraytracer_amd/librtr_hip.so" "$ROOT/realtimeraytracer_amd/librtr_hip_$NAME.so"
rm -rf "$TMP"
echo "built realtimeraytracer_amd/librtr_hip_$NAME.so with $FLAGS"
