#!/bin/bash
# Builds tools/engine_fuzz_asan.cpp with the whole library, host code under AddressSanitizer (device code as in the product).
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
S=opus_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-strict-aliasing -w \
  -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
  -o variants/engine_fuzz_asan tools/engine_fuzz_asan.cpp \
  $S/host/packet.cpp $S/host/celt_parse.cpp $S/host/silk_parse.cpp $S/host/stream_parser.cpp $S/host/capi_parse.cpp $S/host/leaf_sort.cpp \
  $S/host/container.cpp $S/host/celt_tables.cpp $S/host/silk_tables.cpp $S/host/engine.cpp $S/hip/celt_launch.hip $S/hip/silk_launch.hip -lpthread
echo variants/engine_fuzz_asan
