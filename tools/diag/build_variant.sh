#!/bin/bash
# Build df-gnn_amd/libdfgnn_<name>.so = the shipped objects with gt_dense.o recompiled under extra flags.
# usage: tools/diag/build_variant.sh <name> [flags...]   e.g.  build_variant.sh stamps -DDFGNN_STAMPS
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/df-gnn_amd/csrc
mkdir -p $root/build/variants/$name
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics"
objs=""
for f in gt_dense ${VARIANT_FILES:-}; do
  /opt/rocm/bin/hipcc $flags "$@" -c $cs/$f.hip -o $root/build/variants/$name/$f.o &
done
wait
for o in $cs/*.o; do
  b=$(basename $o .o)
  if [ -f $root/build/variants/$name/$b.o ]; then objs="$objs $root/build/variants/$name/$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/df-gnn_amd/libdfgnn_$name.so
echo built $root/df-gnn_amd/libdfgnn_$name.so
