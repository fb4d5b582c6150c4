#!/bin/bash
# Build a variant of libfneus_hip.so with extra compiler flags (timing experiments):
#   tools/experiments/build_variant.sh NAME "-DFNEUS_DBG_NO_ROWSTORE" [file.hip ...]
# Only the listed sources (default: sdf_kernels.hip) are recompiled with the flags; the rest is taken from csrc/build.
# Result: factored-neus_amd/fneus/variants/libfneus_NAME.so  (select it with FNEUS_LIB=...)
set -e
cd "$(dirname "$0")/../../factored-neus_amd/csrc"
name=$1; flags=$2; shift 2 || true
srcs=${@:-sdf_kernels.hip}
make -j8 >/dev/null
mkdir -p build/$name ../fneus/variants
objs=""
for f in *.hip; do
  o=build/${f%.hip}.o
  for s in $srcs; do
    if [ "$s" == "$f" ]; then
      o=build/$name/${f%.hip}.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-value -I../../include $flags -c $f -o $o
    fi
  done
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../fneus/variants/libfneus_$name.so $objs
echo built ../fneus/variants/libfneus_$name.so
