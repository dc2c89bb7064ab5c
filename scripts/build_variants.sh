#!/bin/bash
# Build container: cross-compile variants of the library for scripts/variant_probe.py.
#   scripts/build_variants.sh name1="<flags>" name2="<flags>" ...   ->  geocalib_amd/lib/variants/<name>.so   (git-ignored; travels with gpurun)
cd $(dirname $0)/..
mkdir -p geocalib_amd/lib/variants
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  rm -rf /tmp/gclm_variant_obj; mkdir -p /tmp/gclm_variant_obj
  make -s -C geocalib_amd/csrc OBJ=/tmp/gclm_variant_obj OUT=/tmp/gclm_variant_obj PASS_FLAGS="$flags" 2>&1 | grep -E "error|warning"
  cp /tmp/gclm_variant_obj/libgeocalib_hip.so geocalib_amd/lib/variants/$name.so && echo "built geocalib_amd/lib/variants/$name.so [$flags]"
done
