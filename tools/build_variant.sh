#!/bin/bash
# Build a measurement variant of the library beside the product one (never shipped, never imported by default):
#   tools/build_variant.sh NAME FILE.hip "-DFLAG=.. -DFLAG2"   ->  tools/ab/lib_NAME.so
# The variant = the product objects (the Makefile's SRCS, nothing else that may lie in csrc/) with FILE.hip recompiled
# under the extra flags; the compiler's messages go to tools/ab/NAME.log and, on failure, to the terminal.  Use it with
#   GPITCH_AMD_LIB=tools/ab/lib_NAME.so python tools/<probe>.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; FLAGS=$3
CS=$ROOT/gpitch_amd/csrc
mkdir -p $ROOT/tools/ab
make -s -C $CS >/dev/null
VOBJ=$ROOT/tools/ab/${NAME}_${FILE%.hip}.o
if ! /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on $FLAGS -c $CS/$FILE -o $VOBJ > $ROOT/tools/ab/$NAME.log 2>&1; then
  cat $ROOT/tools/ab/$NAME.log; exit 1
fi
SRCS=$(sed -n 's/^SRCS *= *//p' $CS/Makefile)
OBJS=""
for src in $SRCS; do
  if [ "$src" == "$FILE" ]; then OBJS="$OBJS $VOBJ"; else OBJS="$OBJS $CS/${src%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/lib_$NAME.so $OBJS
echo "built tools/ab/lib_$NAME.so"
