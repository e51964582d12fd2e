#!/bin/bash
# Build libnimg.so for gfx950 (MI355X) in-tree.  Usage: neural-imaging_amd/csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
SRCS=$(ls *.hip)
OUT=../libnimg.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function \
      -I../../include $SRCS -o $OUT "$@"
echo "built $(realpath $OUT)"
