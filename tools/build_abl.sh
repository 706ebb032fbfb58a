#!/bin/bash
# A/B builds of libvcx.so with one compile-time switch changed (git-ignored, tools/_abl/): same sources, same flags, same ABI.
#   tools/build_abl.sh gnold -DVCX_GN_TWO_PHASE      -> tools/_abl/libvcx_gnold.so     (the switches: csrc/vcx_ablate.h)
set -e
name=$1; shift
cd "$(dirname "$0")/../viewcrafter_amd/csrc"
out=../../tools/_abl; mkdir -p $out/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wall -Wno-unused-function"
for f in api gemm gemm_dma gemm_ws attention attention_v2 norm elementwise; do
    extra=""; [ $f = attention_v2 ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c $f.hip -o $out/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libvcx_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
ls -la $out/libvcx_$name.so
