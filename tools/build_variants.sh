#!/bin/bash
# Build kernel variants (extra -D flags) HERE, in parallel, into aten_amd/_variants/ (the .so files travel to the GPU box
# with the snapshot; *.so is git-ignored).   usage: tools/build_variants.sh "name:flags" ...
cd "$(dirname "$0")/.."
mkdir -p aten_amd/_variants
J=${J:-4}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  [ "$flags" = "$v" ] && flags=""
  bid=$(python -c "import sys; from aten_amd.build import build_id; print(build_id(sys.argv[1:]))" $flags)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
      -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result -I include $flags "-DATN_BUILD_ID=\"$bid\"" \
      -o aten_amd/_variants/libaten_amd_$name.so aten_amd/csrc/aten_amd.hip > aten_amd/_variants/$name.build.log 2>&1 \
      && echo "built $name" || { echo "$name: BUILD FAILED"; tail -5 aten_amd/_variants/$name.build.log; } ) &
  while [ "$(jobs -rp | wc -l)" -ge "$J" ]; do sleep 1; done
done
wait
