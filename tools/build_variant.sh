#!/bin/bash
# Builds the tree's sources (or those of a git revision: -r REV) into semantic-search_amd/lib/ab/<name>.so for
# tools/ab.sh: tools/build_variant.sh [-r REV] name [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REV=""
if [ "$1" = "-r" ]; then REV=$2; shift 2; fi
N=$1; shift
SRC=$R/semantic-search_amd/csrc; INC=$R/include
if [ -n "$REV" ]; then
  T=$(mktemp -d); git -C $R archive $REV semantic-search_amd/csrc include | tar -x -C $T
  SRC=$T/semantic-search_amd/csrc; INC=$T/include
fi
O=$(mktemp -d); mkdir -p $R/semantic-search_amd/lib/ab
C="-x hip -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$INC -I$SRC -Wno-unused-result -Wno-unused-value"
for f in geometry label ego engine; do /opt/rocm/bin/hipcc $C -ffp-contract=off "$@" -c $SRC/$f.* -o $O/$f.o & done
/opt/rocm/bin/hipcc $C "$@" -c $SRC/fuse.hip -o $O/fuse.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/semantic-search_amd/lib/ab/$N.so $O/*.o
rm -rf $O; echo built $N
