#!/bin/bash
# Host-side sanitizer builds of the library and the engine stress tool (device code is compiled as usual:
# sanitizers for GPU code are not available on the pool these run on).
#   tools/sanitize_host.sh address   -> build/asan/{liboval_hip.so,engine_stress}   (AddressSanitizer + UBSan)
#   tools/sanitize_host.sh thread    -> build/tsan/...   (ThreadSanitizer: links, but the HIP runtime finds no device under
#                                        that sanitizer's address-space layout, so there is nothing to run with it)
# Run on a GPU box as:  ASAN_OPTIONS=detect_leaks=0 build/asan/engine_stress <seconds> <threads>
set -euo pipefail
kind=${1:-address}
case "$kind" in
  address) dir=build/asan; san="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer" ;;
  thread)  dir=build/tsan; san="-Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer" ;;
  *) echo "usage: $0 address|thread" >&2; exit 2 ;;
esac
cd "$(dirname "$0")/.."
mkdir -p "$dir"
src=oval_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC $san -x hip \
  $src/oval_kernels.hip $src/oval_capi.cpp $src/oval_pattern.cpp $src/oval_engine.cpp $src/oval_context.cpp $src/oval_comm.cpp \
  -shared -ldl -o "$dir/liboval_hip.so"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 $san -o "$dir/engine_stress" tools/engine_stress.cpp -I$src \
  -L"$dir" -loval_hip -Wl,-rpath,'$ORIGIN' -lpthread
ls -la "$dir"
