#!/bin/bash
# Development aid: build jpeg-decoder_amd/build/var_<name>.so with extra -D flags for A/B timing
# (select with JPEGAMD_LIB=...).  usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../jpeg-decoder_amd"
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result -I../include -Icsrc "$@" -c csrc/jdgpu.hip -o build/var_$name.o
gcc -O2 -std=c11 -fPIC -I../include -Icsrc "$@" -c csrc/jdgpu_prep.c -o build/var_${name}_prep.o
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 build/var_$name.o build/var_${name}_prep.o build/jpeg_host.o -pthread -o build/var_$name.so
echo built build/var_$name.so
