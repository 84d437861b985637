#!/bin/bash
# Build a variant of the library that differs from the default build in the compile-time macros of ONE translation unit:
#   bash tools/build_variant.sh NAME tv "-DSBTV_DIET=0"      -> semi-blind-.../lib/libsbtv_NAME.so
# (the other objects are those of the default build: run `make -C .../csrc` first).  A/B such libraries on the GPU box with
# tools/ab_lib.py (SBTV_LIBRARY selects the library a process loads).
set -eo pipefail
NAME=${1:?name}; UNIT=${2:?translation unit without .hip}; FLAGS=${3:-}
R=$(cd "$(dirname "$0")/.." && pwd)
C="$R/semi-blind-image-deblurring-problems-with-tv_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
cd "$C"
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc $FLAGS -c $UNIT.hip -o $UNIT.$NAME.o
OBJS=""
for u in ctx tv fft elementwise salsa sapg admm group; do
  if [ "$u" = "$UNIT" ]; then OBJS="$OBJS $UNIT.$NAME.o"; else OBJS="$OBJS $u.o"; fi
done
$HIPCC -shared -fPIC --offload-arch=gfx950 -o ../lib/libsbtv_$NAME.so $OBJS -lpthread
echo "built ../lib/libsbtv_$NAME.so"
