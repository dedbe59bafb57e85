#!/bin/bash
export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
# Variants of libnautilus_hip.so that differ in the branch-and-bound matcher's compile-time switches, into
# build/variants/ (run on the GPU box through NHIP_LIB: tools/bnb_quick.py, tools/bnb_ab.py).
#   tools/bnb_variants.sh name1 "-Dflags1" name2 "-Dflags2" ...
set -e
cd "$(dirname "$0")/../nautilus_amd/csrc"
make -s -j4
OUT=../../build/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off"
build() {
  name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS $@ -c nhip_bnb.hip -o $OUT/bnb_$name.o
  /opt/rocm/bin/hipcc $FLAGS $@ -c nhip_bnb_instr.hip -o $OUT/bnbi_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libbnb_$name.so nhip_api.o nhip_grid.o nhip_csm.o nhip_csm16.o nhip_csm_small.o $OUT/bnb_$name.o $OUT/bnbi_$name.o nhip_lc.o nhip_resid.o nhip_corr.o -ldl
  rm -f $OUT/bnb_$name.o $OUT/bnbi_$name.o
}
while [ $# -gt 0 ]; do
  build "$1" $2 &
  shift; shift
done
wait
ls $OUT
