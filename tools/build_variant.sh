#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-DKNOB=value ...]
# builds tools/bin/lib_<name>.so = the library with <file.hip> recompiled under the extra flags (the other objects are
# taken from the in-tree build), for same-box A/B runs (tools/ab_bench.sh, tools/ab_libs.py)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../medicalimageanalysis_amd/csrc"
make -s >/dev/null
base=$(basename $src .hip)
extra=""
case $base in vmf_general|vmf3_fast|consumers) extra="-fno-slp-vectorize";; esac
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -w $extra "$@" -c $base.hip -o /tmp/variant_$name.o
objs=""
for o in vmf3 vmf3_fast elementwise vmf_general consumers; do
  if [ $o = $base ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/lib_$name.so $objs -ldl
echo built tools/bin/lib_$name.so
