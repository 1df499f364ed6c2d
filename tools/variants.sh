#!/bin/bash
# Build kernel variants of libsmafa_amd.so side by side (smafa_amd/lib_v<N>/) for A/B runs on the GPU box:
#   tools/variants.sh "-DSMAFA_ZONE_VARIANT=1" "-DSMAFA_ZONE_VARIANT=2" ...      (here, cross-compiling)
# then on the box:  SMAFA_AMD_LIB=smafa_amd/lib_v1/libsmafa_amd.so python bench.py ...
set -e
cd "$(dirname "$0")/../smafa_amd/csrc"
n=1
for flags in "$@"; do
  make -j8 LIBDIR=../lib_v$n OBJDIR=../build_v$n BINDIR=../bin_v$n EXTRA_HIPFLAGS="$flags" ../lib_v$n/libsmafa_amd.so > /dev/null
  echo "lib_v$n: $flags"
  n=$((n+1))
done
