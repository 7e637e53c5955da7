#!/bin/bash
# development: build_abl/<name>.so = HEAD's objects with ONE source recompiled under extra flags (cross-compiles here, no GPU)
#   tools/dev/mkvariant.sh <name> <source.hip> "<extra flags>" ["<replacement per-source flags>"]
set -e
name=$1; src=$2; extra=$3
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form"
per=$(python - <<PY
import __graft_entry__ as g
print(" ".join(g.SOURCE_FLAGS.get("$src", [])))
PY
)
if [ -n "${4+x}" ]; then per=$4; fi
mkdir -p build_abl/obj_$name
/opt/rocm/bin/hipcc $base $per $extra -c oatomobile_amd/csrc/$src -o build_abl/obj_$name/$src.o
objs=""
for f in $(python -c "import __graft_entry__ as g; print(' '.join(g._sources()))"); do
  if [ "$f" == "$src" ]; then objs="$objs build_abl/obj_$name/$src.o"; else objs="$objs build/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o build_abl/$name.so
echo "built build_abl/$name.so"
