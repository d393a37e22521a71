#!/bin/bash
# ASan + UBSan over everything that runs on the CPU: the oracle (oracle/aten_oracle.cpp) and the product's host library
# (aten_amd/csrc/host/*.cpp), exercised by the whole `-m "not gpu"` suite.  The in-tree .so files are swapped for
# instrumented builds and restored afterwards.  usage: tools/sanitize_cpu.sh   (prints every "runtime error" / ASan report)
set -u
cd "$(dirname "$0")/.."
cp oracle/liboracle.so /tmp/liboracle_backup.so
cp aten_amd/libaten_amd_scene.so /tmp/libaten_amd_scene_backup.so
restore() { cp /tmp/liboracle_backup.so oracle/liboracle.so; cp /tmp/libaten_amd_scene_backup.so aten_amd/libaten_amd_scene.so; }
trap restore EXIT
SAN="-O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fsanitize=undefined,address -shared"
g++ $SAN -fopenmp -o oracle/liboracle.so oracle/aten_oracle.cpp || exit 1
g++ $SAN -o aten_amd/libaten_amd_scene.so aten_amd/csrc/host/bvh_builder.cpp aten_amd/csrc/host/camera.cpp aten_amd/csrc/host/obj_ingest.cpp || exit 1
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    timeout 2400 python -m pytest tests -q -m "not gpu" -p no:cacheprovider -s > /tmp/sanitize_cpu.log 2>&1
tail -1 /tmp/sanitize_cpu.log
grep -E "runtime error|AddressSanitizer" /tmp/sanitize_cpu.log | sort | uniq -c
echo "reports: $(grep -cE 'runtime error|AddressSanitizer' /tmp/sanitize_cpu.log)"
