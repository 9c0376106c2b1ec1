#!/bin/bash
# A variant of the library for an interleaved A/B (tools/ab_variants.sh, tools/r5_ab_lib.sh, tools/sessions/*):
#   tools/build_variant.sh NAME FILE.hip -DFLAG[=V] ...   ->  jpeg_gpu_amd/variants/NAME.so
# = the tuning build's objects with FILE.hip recompiled under the flags (and its ISA listing kept beside it).
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
python -m jpeg_gpu_amd.build > /dev/null
B=jpeg_gpu_amd/build; V=jpeg_gpu_amd/variants; mkdir -p $V $B/variant_$NAME
hipcc -save-temps=obj --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall \
  "$@" -c jpeg_gpu_amd/csrc/$SRC -o $B/variant_$NAME/$SRC.o
OBJS=""
for o in layout.c.tuning entropy.c libjpeg_vtbl.c band.c device_api.cpp vtbl.cpp pipeline.cpp huff_prepare.cpp huff_api.cpp \
         idct_kernels.hip huff_kernels.hip pack_kernels.hip unstuff_kernels.hip copy_kernel.hip; do
  if [ "$o" = "$SRC" ]; then OBJS="$OBJS $B/variant_$NAME/$SRC.o"; else OBJS="$OBJS $B/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$NAME.so $OBJS -lpthread -ldl
echo $V/$NAME.so
