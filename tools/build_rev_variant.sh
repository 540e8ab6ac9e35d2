#!/bin/bash
# Build libneuronika_hip.so of ANOTHER git revision as a variant library for same-box A/B runs:
#   bash tools/build_rev_variant.sh r02 ae2101c   ->  benchmarks/_ab/r02.so   (use: NEURONIKA_HIP_LIB=benchmarks/_ab/r02.so ...)
# (only the raw C ABI is comparable across revisions: the C++ tape is linked against the in-tree library)
set -eu
name=$1; rev=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" neuronika_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/benchmarks/_ab/obj_$name"
objs=()
for src in "$tmp"/neuronika_amd/csrc/*.hip; do
  o="$root/benchmarks/_ab/obj_$name/$(basename "${src%.hip}").o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$tmp/include" -I"$tmp/neuronika_amd/csrc" -c "$src" -o "$o" 2>/dev/null &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/benchmarks/_ab/$name.so" "${objs[@]}" -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$tmp"
echo "$root/benchmarks/_ab/$name.so"
