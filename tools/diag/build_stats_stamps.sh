#!/bin/bash
# Build df-gnn_amd/libdfgnn_sstamps.so = the shipped objects with gt_dense_stats.o recompiled with phase stamps
# (-DDFGNN_STAMPS -DDFGNN_STAMPS_TU: that translation unit then owns the stamp variables and their setters).
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/df-gnn_amd/csrc
mkdir -p $root/build/variants/sstamps
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $flags -DDFGNN_STAMPS -DDFGNN_STAMPS_TU "$@" -c $cs/gt_dense_stats.hip -o $root/build/variants/sstamps/gt_dense_stats.o
objs=""
for o in $cs/*.o; do
  b=$(basename $o .o)
  if [ "$b" = gt_dense_stats ]; then objs="$objs $root/build/variants/sstamps/gt_dense_stats.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/df-gnn_amd/libdfgnn_sstamps.so
echo built $root/df-gnn_amd/libdfgnn_sstamps.so
