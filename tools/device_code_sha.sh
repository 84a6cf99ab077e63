#!/bin/bash
# sha256 of the gfx950 DEVICE code object of csrc/mapalign.hip at a commit (default: the working tree).  Host-only edits
# of mapalign.hip change bench.py's kernel_source_sha (a hash of the source files) but not this: two commits with the same
# value here run the same kernels, instruction for instruction, and the rocprofv3 summaries of one hold for the other.
# (Compiled from a scratch directory with relative paths only: hipcc derives its compilation-unit id from the options.)
#   bash tools/device_code_sha.sh            # working tree
#   bash tools/device_code_sha.sh 6ce7cc8    # a commit
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
trap 'rm -rf "$D"' EXIT
if [ $# -ge 1 ]; then
  for f in $(git -C "$R" ls-tree --name-only "$1" 3d-localisation-and-mapping_amd/csrc/); do
    git -C "$R" show "$1:$f" > "$D/$(basename "$f")"
  done
  git -C "$R" show "$1:include/mapalign.h" > "$D/mapalign.h"
else
  cp "$R"/3d-localisation-and-mapping_amd/csrc/* "$D"/
  cp "$R/include/mapalign.h" "$D"/
fi
cd "$D"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. --cuda-device-only -c -o dev.o mapalign.hip
sha256sum dev.o | cut -c1-16
