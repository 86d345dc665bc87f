#!/bin/bash
# Build lib/libvitx_<name>.so = the current objects with the listed sources recompiled under extra flags (same-box A/B experiments):
#   tools/build_variant.sh prio1 "-DVITX_XP_SETPRIO=1" gemm_bf16_pipe.hip [more.hip ...]
set -e
cd "$(dirname "$0")/../vit-tensorflow_amd"
name=$1; flags=$2; shift 2
python build.py > /dev/null
mkdir -p build_variant/$name
objs=$(ls build/*.o)
for src in "$@"; do
  o=build_variant/$name/${src%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c csrc/$src -o $o
  objs=$(echo "$objs" | grep -v "build/${src%.hip}.o"); objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libvitx_$name.so $objs -ldl
echo lib/libvitx_$name.so
