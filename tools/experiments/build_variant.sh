#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [...]: glenet_amd/csrc/libglenet_hip_<name>.so = the library
# with the listed sources recompiled under the extra flags (the other objects as built); run with GLX_HIP_LIB=<that path>.
set -e
cd "$(dirname "$0")/../glenet_amd/csrc"
NAME=$1; EXTRA=$2; shift 2
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable"
OBJS=""
for f in *.hip; do
  o=${f%.hip}.o
  for v in "$@"; do
    if [ "$v" == "$f" ]; then
      o=${f%.hip}_$NAME.vo
      /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f -o $o
    fi
  done
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libglenet_hip_$NAME.so $OBJS
echo $(pwd)/libglenet_hip_$NAME.so
