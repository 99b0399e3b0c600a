#!/bin/bash
# Build libkwy.so (gfx950 HIP kernels + C ABI) in-tree: kwiiyatta_amd/libkwy.so
set -e
cd "$(dirname "$0")"
OUT=../libkwy.so
SRCS=$(ls *.hip)
mkdir -p obj
PIDS=()
for f in $SRCS; do
  o=obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ kwy_device.hpp -nt "$o" ] || [ kwy_internal.hpp -nt "$o" ] || [ ../../include/kwy.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -c "$f" -o "$o" &
    PIDS+=($!)
  fi
done
for p in "${PIDS[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT obj/*.o
echo "built $OUT"
# test-only helper library (device-side building blocks on caller data); not loaded by the package
if [ ! -f ../libkwy_selftest.so ] || [ selftest/kwy_selftest.hip -nt ../libkwy_selftest.so ] || [ kwy_device.hpp -nt ../libkwy_selftest.so ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -o ../libkwy_selftest.so selftest/kwy_selftest.hip
  echo "built ../libkwy_selftest.so"
fi
