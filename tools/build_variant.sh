#!/bin/bash
# Build gshell_amd/lib/variants/<name>.so = the library with one source recompiled under extra -D flags (no GPU needed; the .so
# travels to the GPU box with gpurun).  Use with GSHELL_HIP_LIB=gshell_amd/lib/variants/<name>.so.
# usage: tools/build_variant.sh <name> <source.hip> -DFOO=1 [-DBAR=2 ...]
set -e
cd "$(dirname "$0")/.."
name="$1"; src="$2"; shift 2
mkdir -p gshell_amd/lib/variants /tmp/gs_variant_$name
objs=""
for o in gshell_amd/lib/obj/*.o; do
  b=$(basename "$o" .o)
  if [ "$b.hip" = "$src" ]; then
    fileflags=$(sed -n 's|^// GS_CXXFLAGS: ||p' "gshell_amd/csrc/$src")       # the source's own per-file flags (csrc/Makefile); GS_NO_FILEFLAGS=1 drops them
    [ -n "$GS_NO_FILEFLAGS" ] && fileflags=""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result $fileflags "$@" \
      -c "gshell_amd/csrc/$src" -o "/tmp/gs_variant_$name/$b.o"
    objs="$objs /tmp/gs_variant_$name/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "gshell_amd/lib/variants/$name.so" $objs
echo "gshell_amd/lib/variants/$name.so"
