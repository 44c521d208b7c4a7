#!/bin/bash
# Build a measurement variant of the library beside the product one (never shipped, never imported by default):
#   tools/build_variant.sh NAME FILE.hip "-DFLAG=.. -DFLAG2"   ->  tools/ab/lib_NAME.so
# The variant = the product objects with FILE.hip recompiled under the extra flags.  Use it with
#   GPITCH_AMD_LIB=tools/ab/lib_NAME.so python tools/<probe>.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; FLAGS=$3
CS=$ROOT/gpitch_amd/csrc
mkdir -p $ROOT/tools/ab
make -s -C $CS >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on $FLAGS -c $CS/$FILE -o $ROOT/tools/ab/${NAME}_${FILE%.hip}.o
OBJS=""
for f in $CS/*.o; do
  if [ "$(basename $f)" == "${FILE%.hip}.o" ]; then OBJS="$OBJS $ROOT/tools/ab/${NAME}_${FILE%.hip}.o"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/lib_$NAME.so $OBJS
echo "built tools/ab/lib_$NAME.so"
