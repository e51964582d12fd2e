#!/bin/bash
# A/B build of libnimg: recompile only the named translation units with extra flags and link them with the product objects.
#   tools/build_variant.sh <tag> "<extra hipcc flags>" conv_bf16 [wgrad5 ...]   ->  neural-imaging_amd/libnimg_<tag>.so
set -e
TAG=$1; EXTRA=$2; shift 2
cd "$(dirname "$0")/../neural-imaging_amd/csrc"
make -j8 > /dev/null
mkdir -p obj_ab
OBJS=""
for o in obj/*.o; do
  b=$(basename $o .o); use=$o
  for u in "$@"; do
    if [ "$u" == "$b" ]; then
      fl="-ffp-contract=fast"; [ "$b" == "djpeg" -o "$b" == "datafeed" ] && fl="-ffp-contract=off"
      [ "$b" == "frontend" ] && fl="$fl -fno-slp-vectorize"
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include $fl $EXTRA -c $b.hip -o obj_ab/${b}_$TAG.o 2> /dev/null
      use=obj_ab/${b}_$TAG.o
    fi
  done
  OBJS="$OBJS $use"
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../libnimg_$TAG.so
echo "built neural-imaging_amd/libnimg_$TAG.so"
