#!/bin/bash
# Build df-gnn_amd/libdfgnn_<name>.so = the shipped objects with gt_dense_stats_w.o recompiled under extra flags.
# usage: tools/diag/build_w_variant.sh <name> [flags...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/df-gnn_amd/csrc
mkdir -p $root/build/variants/$name
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $flags "$@" -c $cs/gt_dense_stats_w.hip -o $root/build/variants/$name/gt_dense_stats_w.o
objs=""
for o in $cs/*.o; do
  b=$(basename $o .o)
  if [ "$b" = gt_dense_stats_w ]; then objs="$objs $root/build/variants/$name/gt_dense_stats_w.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic $objs -o $root/df-gnn_amd/libdfgnn_$name.so
echo built $root/df-gnn_amd/libdfgnn_$name.so
