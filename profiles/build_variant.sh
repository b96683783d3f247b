#!/bin/bash
# Builds librtr_hip.so of another git revision next to the working tree's, for same-box A/B runs with profiles/ab_lib.sh:
#   bash profiles/build_variant.sh <git-rev> <name>     ->  realtimeraytracer_amd/librtr_hip_<name>.so
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/rtr_variant_XXXX)
git -C "$ROOT" archive "$REV" include realtimeraytracer_amd/csrc | tar -x -C "$TMP"
make -C "$TMP/realtimeraytracer_amd/csrc" -j4 ../librtr_hip.so > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
cp "$TMP/realtimeraytracer_amd/librtr_hip.so" "$ROOT/realtimeraytracer_amd/librtr_hip_$NAME.so"
rm -rf "$TMP"
echo "built realtimeraytracer_amd/librtr_hip_$NAME.so from $REV"
