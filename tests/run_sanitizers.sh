#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side native code (the host scene layer behind librtr_host.so and the oracle), driven by the
# CPU tests that exercise them.  GPU sanitizers are not available on the pool; this is the CPU build only.
#   bash tests/run_sanitizers.sh        (from the repo root; restores the normal libraries afterwards)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
cp realtimeraytracer_amd/librtr_host.so /tmp/librtr_host.so.orig
cp oracle/liboracle.so /tmp/liboracle.so.orig
restore() { cp /tmp/librtr_host.so.orig realtimeraytracer_amd/librtr_host.so; cp /tmp/liboracle.so.orig oracle/liboracle.so; }
trap restore EXIT
g++ $SAN -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Irealtimeraytracer_amd/csrc -Iinclude -shared -o realtimeraytracer_amd/librtr_host.so realtimeraytracer_amd/csrc/host/rtr_host_api.cpp
g++ $SAN -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -Iinclude -shared -o oracle/liboracle.so oracle/oracle_render.cpp oracle/oracle_post.cpp
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_host_scene.py tests/test_obj_ingest.py tests/test_images.py tests/test_oracle_bvh.py tests/test_ltc_tables.py tests/test_witness.py -x -q -m "not gpu" "$@"
