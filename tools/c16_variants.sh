#!/bin/bash
export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
# Builds variants of libnautilus_hip.so that differ in csm_correlate16_kernel's compile-time shape (tile rows, waves
# per workgroup, SDWA adds) into build/variants/, for tools/c16_time.py (run on the GPU box through NHIP_LIB).
set -e
cd "$(dirname "$0")/../nautilus_amd/csrc"
make -s -j4
OUT=../../build/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off"
build() {  # name, defines...
  name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c nhip_csm16.hip -o $OUT/c16_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so nhip_api.o nhip_grid.o nhip_csm.o $OUT/c16_$name.o nhip_bnb.o nhip_bnb_instr.o nhip_lc.o nhip_resid.o nhip_corr.o -ldl
  rm -f $OUT/c16_$name.o
}
rm -f $OUT/lib_*.so
build base &
build w4r120 -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=120 -DNHIP_C16_WAVES_PER_SIMD=3 &
build w4r96 -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=96 -DNHIP_C16_WAVES_PER_SIMD=4 &
build w4r96s -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=96 -DNHIP_C16_WAVES_PER_SIMD=4 -DNHIP_C16_SDWA=1 &
wait
build w4r104 -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=104 -DNHIP_C16_WAVES_PER_SIMD=3 &
build w4r112 -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=112 -DNHIP_C16_WAVES_PER_SIMD=3 &
build w4r120f8 -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=120 -DNHIP_C16_WAVES_PER_SIMD=3 -DNHIP_C16_FILL_INFLIGHT=8 &
build w4r120s -DNHIP_C16_WG_WAVES=4 -DNHIP_C16_TILE_ROWS=120 -DNHIP_C16_WAVES_PER_SIMD=3 -DNHIP_C16_SDWA=1 &
wait
ls -la $OUT
