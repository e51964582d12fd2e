#!/bin/bash
# Build libnimg.so for gfx950 (MI355X) in-tree.  Usage: neural-imaging_amd/csrc/build.sh [make variables, e.g. EXTRA=-DX]
set -e
cd "$(dirname "$0")"
make -j"$(nproc)" "$@"
