#!/bin/bash
# Build the library of an OLDER COMMIT into geocalib_amd/lib/ab/libgeocalib_hip_<tag>.so for same-box A/B runs
# (scripts/ab_lib.sh).  usage: scripts/build_ref_lib.sh <commit> <tag>      e.g. scripts/build_ref_lib.sh 2d28bf8 r01
set -e
COMMIT=${1:?commit}; TAG=${2:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/geocalib_amd/csrc $TMP/include $TMP/geocalib_amd/lib $ROOT/geocalib_amd/lib/ab
for f in $(git -C $ROOT ls-tree --name-only $COMMIT geocalib_amd/csrc/); do git -C $ROOT show $COMMIT:$f > $TMP/$f; done
git -C $ROOT show $COMMIT:include/gclm.h > $TMP/include/gclm.h
make -C $TMP/geocalib_amd/csrc -j4 > /dev/null
cp $TMP/geocalib_amd/lib/libgeocalib_hip.so $ROOT/geocalib_amd/lib/ab/libgeocalib_hip_$TAG.so
rm -rf $TMP
echo "built geocalib_amd/lib/ab/libgeocalib_hip_$TAG.so from $COMMIT"
