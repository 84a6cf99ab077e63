#!/bin/bash
# sha256 of the gfx950 DEVICE code of each kernel translation unit of csrc/ at a commit (default: the working tree), and
# one hash over all of them.  Host-only edits change bench.py's kernel_source_sha (a hash of source files) but not this:
# two commits with the same values here run the same kernels, instruction for instruction, and the rocprofv3 summaries of
# one hold for the other.  (Compiled from a scratch directory with relative paths only: hipcc derives its
# compilation-unit id from the options.)
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
UNITS=$(ls kernels_*.hip 2>/dev/null || true)
[ -n "$UNITS" ] || UNITS=mapalign.hip          # commits before the split (round 4 and earlier)
for u in $UNITS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -mllvm -amdgpu-kernarg-preload-count=16 \
      --cuda-device-only -c -o "${u%.hip}.o" "$u" &
done
wait
for u in $UNITS; do echo "$(sha256sum "${u%.hip}.o" | cut -c1-16)  $u"; done
cat $(for u in $UNITS; do echo "${u%.hip}.o"; done) | sha256sum | cut -c1-16
