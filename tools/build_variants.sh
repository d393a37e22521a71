#!/bin/bash
# Build kernel variants (extra compiler flags on every translation unit) HERE, in parallel, into aten_amd/_variants/ (the .so
# files travel to the GPU box with the snapshot; *.so is git-ignored).   usage: tools/build_variants.sh "name:flags" ...
# A flag of the form unit=FLAG is not supported: per-unit flags are aten_amd.build.HIP_UNITS.
cd "$(dirname "$0")/.."
mkdir -p aten_amd/_variants
J=${J:-3}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  [ "$flags" = "$v" ] && flags=""
  ( python -c "
import sys
from aten_amd import build as b
b.hip_compile('aten_amd/_variants/libaten_amd_$name.so', sys.argv[1:], objdir='aten_amd/_variants/_obj')" $flags > aten_amd/_variants/$name.build.log 2>&1 \
      && echo "built $name" || { echo "$name: BUILD FAILED"; tail -5 aten_amd/_variants/$name.build.log; } ) &
  while [ "$(jobs -rp | wc -l)" -ge "$J" ]; do sleep 1; done
done
wait
