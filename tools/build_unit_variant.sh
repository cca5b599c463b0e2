#!/bin/bash
# Diagnostic library with ONE unit rebuilt with extra flags, the rest taken from the production objects (klhr_amd/lib/obj):
#   tools/build_unit_variant.sh NAME UNIT [flags...]     UNIT: hot_v0_o0 | split_v0 | ...   ->  klhr_amd/lib/diag/libklhr_NAME.so
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/klhr_amd/lib/obj
OUT=$ROOT/klhr_amd/lib/diag/$NAME
mkdir -p $OUT
case $UNIT in
  hot_v*) V=${UNIT:5:1}; O=${UNIT:8:1}; SRC="$ROOT/klhr_amd/csrc/klhr_hot.hip -DKLHR_VARIANT=$V -DKLHR_OVERRELAX=$O";;
  split_v*) V=${UNIT:7:1}; SRC="$ROOT/klhr_amd/csrc/klhr_split.hip -DKLHR_VARIANT=$V";;
  *) echo "unknown unit $UNIT"; exit 1;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c $SRC -o $OUT/$UNIT.o
OBJS=""
for d in api hot_v0_o0 hot_v0_o1 hot_v1_o0 hot_v1_o1 hot_v2_o0 hot_v2_o1 hot_v3_o0 split_v0 split_v1 split_v2; do
  if [ $d = $UNIT ]; then OBJS="$OBJS $OUT/$UNIT.o"; else OBJS="$OBJS $OBJ/$d/$d.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/klhr_amd/lib/diag/libklhr_$NAME.so $OBJS
ls -la $ROOT/klhr_amd/lib/diag/libklhr_$NAME.so
