#!/bin/bash
# Diagnostic builds of the library with one source file recompiled under extra macros (ablations give WRONG results):
#   tools/_diag/build_variants.sh conv3x3.hip NAME "-DKA_DIAG -DKA_PC_NO_W" [NAME2 "flags2" ...]  ->  keisei_amd/libka_NAME.so
set -e
cd "$(dirname "$0")/../.."
src=$1; shift
python -m keisei_amd.build > /dev/null
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  obj=keisei_amd/csrc/_build/$src.$name.o
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-unknown-pragmas $flags -c keisei_amd/csrc/$src -o $obj 2>/dev/null
  objs=$(ls keisei_amd/csrc/_build/*.hip.o | grep -v "/$src.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o keisei_amd/libka_$name.so $objs $obj
  echo "built keisei_amd/libka_$name.so ($flags)"
done
