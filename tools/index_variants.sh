#!/bin/bash
# On the GPU box: tools/index_ab.py for the default library and every smafa_amd/lib_v*/ build (kernel variants of the probe).
cd "$(dirname "$0")/.."
echo "== base"; python3 tools/index_ab.py "$@" 2>&1 | grep "index:"
for lib in smafa_amd/lib_v*/libsmafa_amd.so; do
  [ -f "$lib" ] || continue
  echo "== $(basename $(dirname $lib))"; SMAFA_AMD_LIB=$PWD/$lib python3 tools/index_ab.py "$@" 2>&1 | grep "index:"
done
