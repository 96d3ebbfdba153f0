#!/bin/bash
# A/B aid: build a variant libarx into a-recsys_amd/arx/lib/exp/<name>.so with extra -D flags
# (select with ARX_LIB=<path> at run time).  usage: tools/build_variant.sh <name> -DARX_WIN_NB1=4 ...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=/tmp/arx_variant_$NAME
mkdir -p $OBJ $ROOT/a-recsys_amd/arx/lib/exp
cd $ROOT/a-recsys_amd/csrc
for f in *.hip; do
  o=$OBJ/${f%.hip}.o
  # only optim*.hip depend on the K7 macros; reuse the main build's objects for the rest
  if [[ $f == ${VARIANT_FILES:-optim}* || ! -f $ROOT/a-recsys_amd/build/${f%.hip}.o ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wno-unused-result "$@" -c $f -o $o &
  else
    cp $ROOT/a-recsys_amd/build/${f%.hip}.o $o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $ROOT/a-recsys_amd/arx/lib/exp/$NAME.so
echo built $ROOT/a-recsys_amd/arx/lib/exp/$NAME.so
