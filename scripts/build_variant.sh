#!/bin/bash
# A second build of libdflash_hip.so with extra -D flags, for same-box A/B runs (DFL_LIB_PATH):
#   scripts/build_variant.sh <out dir> [-DFLAG=..]...
out=$1; shift
mkdir -p "$out/obj"
cd "$(dirname "$0")/../dflash_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function "$@" -c $f -o "$OLDPWD/$out/obj/${f%.hip}.o" &
done
wait
cd "$OLDPWD"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdflash_hip.so" "$out"/obj/*.o && rm -rf "$out/obj" && ls -la "$out/libdflash_hip.so"
