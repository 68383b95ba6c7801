#!/bin/bash
# A second build of the library with other compile-time settings, beside the product's:
#   bash tools/build_variant.sh NAME "-DXYZ_INV_THREADS=512 -DXYZ_INV_PREFETCH=1" [file.hip ...]
# -> sperr_amd/libsperr_hip_NAME.so (load it with SPERR_HIP_LIB=...).  Only the named sources (default:
# xform.hip) are compiled with the extra flags; the other objects are the product build's.
set -e
name=$1; flags=$2; shift 2
srcs=${@:-xform.hip}
here=$(cd $(dirname $0)/../sperr_amd/csrc && pwd)
make -s -C $here
mkdir -p $here/_build/var_$name
objs=""
for f in xform speck_enc speck_dec speck_mx outlier speck2d engine farm; do
  if echo " $srcs " | grep -q " $f.hip "; then
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value $flags -c $here/$f.hip -o $here/_build/var_$name/$f.o
    objs="$objs $here/_build/var_$name/$f.o"
  else
    objs="$objs $here/_build/$f.o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $here/../libsperr_hip_$name.so $objs
echo built $here/../libsperr_hip_$name.so
